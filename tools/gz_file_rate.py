"""`match` over ONE single-member .gz FASTQ through the host pipeline (device gunzip), as bench.py's file_pipeline.gz does it
(developer tool): python tools/gz_file_rate.py [reads] [keep_dir];  GS_HOST_TRACE=1 prints the stages of every batch"""
import os
import sys
import time
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth, host  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
keep = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="gsb_")
os.makedirs(keep, exist_ok=True)
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
plain = os.path.join(keep, "b_%d.fastq" % n)
gz = plain + ".gz"
if not os.path.exists(gz):
    seq, off = synth.reads_host(db.genomes, n)
    bench._fastq_text(seq, n).tofile(plain)
    bench._write_gz(plain, gz, False, 16)
print(f"{os.path.getsize(gz) / 1e6:.0f} MB of gzip", flush=True)
want = None
for rep in range(4):
    t0 = time.perf_counter()
    t, _, tot = host.match_files(store, [gz])[:3]
    dt = time.perf_counter() - t0
    print(f"match, device gunzip: {dt * 1e3:.1f} ms = {n * 150 / dt / 1e9:.2f} Gbp/s, reads {tot.reads}", flush=True)
    assert want is None or np.array_equal(t, want)
    want = t
