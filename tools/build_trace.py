"""phase times of gs_db_create on the GPU box's host cores (developer tool):  python tools/build_trace.py [genera]
(20 species per genus, 100 kbp each: 25 -> 47 M k-mers, 250 -> 473 M)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GS_BUILD_TRACE"] = "1"
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

for genera in [int(x) for x in sys.argv[1:]] or [25]:
    db = synth.SynthDB(genera=genera, species_per_genus=20)
    t0 = time.time()
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    print("genera %d: %d k-mers, gs_db_create %.2f s" % (genera, db.n_entries, time.time() - t0), flush=True)
    store.close()
    if os.environ.get("GS_TRACE_STRIPED"):  # GS_TRACE_STRIPED=8: the same store as 8 stripes (here: all on device 0)
        t0 = time.time()
        ss = ga.DeviceKMerStore.striped(31, db.kmers, db.value_idx, db.n_values, db.parent_vi, devices=(0,) * int(os.environ["GS_TRACE_STRIPED"]))
        print("genera %d: striped create %.2f s" % (genera, time.time() - t0), flush=True)
        for x in ss:
            x.close()
