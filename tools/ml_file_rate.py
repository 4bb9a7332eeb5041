"""multi-line FASTQ through the file pipeline (developer tool): records found on the device (gs_match_submit_fastq_ml) against the
reference-exact parser on one thread (GS_HOST_ML=0).      python tools/ml_file_rate.py [reads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
db = synth.SynthDB(k=31)
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n, read_len=150)
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gs_ml_%d.fastq" % os.getpid())
t0 = time.time()
s2 = seq.reshape(n, 150)
nl = np.full((n, 1), 10, dtype=np.uint8)
q = np.full((n, 50), ord("F"), dtype=np.uint8)
hdr = np.frombuffer(b"@read/1 length=150", dtype=np.uint8)
rec = np.concatenate([np.broadcast_to(hdr, (n, len(hdr))), nl, s2[:, :60], nl, s2[:, 60:120], nl, s2[:, 120:], nl,
                      np.broadcast_to(np.frombuffer(b"+", dtype=np.uint8), (n, 1)), nl, q, q[:, :10], nl, q, q[:, :10], nl, q[:, :30], nl], axis=1)
rec.tofile(path)
size = os.path.getsize(path)
print("wrote %.2f GB (%d reads, sequence in 3 lines, quality in 3) in %.1f s" % (size / 1e9, n, time.time() - t0), flush=True)
for label, env in (("device record search", None), ("host parser (GS_HOST_ML=0)", "0"), ("device record search", None)):
    if env is None:
        os.environ.pop("GS_HOST_ML", None)
    else:
        os.environ["GS_HOST_ML"] = env
    before = host.stat(0)
    t0 = time.time()
    table, _, tot = host.match_files(store, [path])
    dt = time.time() - t0
    print("%-30s %.2f s  %.2f Gbp/s  %.2f GB/s of file  reads %d  ml chunks %d  classified %d" %
          (label, dt, n * 150 / dt / 1e9, size / dt / 1e9, tot.reads, host.stat(0) - before, int(table[:, 0].sum())), flush=True)
os.unlink(path)
