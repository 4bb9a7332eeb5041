"""save / load times of the native store file (developer tool):  python tools/store_file_rate.py [genera]  (250 -> 473 M k-mers)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

genera = int(sys.argv[1]) if len(sys.argv) > 1 else 25
db = synth.SynthDB(genera=genera, species_per_genus=20)
t0 = time.time()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
t1 = time.time()
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gs_rate_%d.gss" % os.getpid())
store.save(path)
t2 = time.time()
store.close()
size = os.path.getsize(path)
s2 = ga.DeviceKMerStore.load(path)
t3 = time.time()
s2.close()
ss = ga.DeviceKMerStore.load_striped(path, devices=(0,) * 8)
t4 = time.time()
for s in ss:
    s.close()
os.unlink(path)
print("k-mers %d  file %.2f GB  gs_db_create %.1f s  save %.1f s  load %.1f s  load striped x8 %.1f s" %
      (db.n_entries, size / 1e9, t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)
