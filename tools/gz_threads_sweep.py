"""file pipeline rate on ONE gzip FASTQ for several inflating-thread counts (developer tool)"""
import gzip
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n)
blk = seq.tobytes()
d = tempfile.mkdtemp(prefix="gsgz")
gz = os.path.join(d, "reads.fastq.gz")
with gzip.open(gz, "wb", compresslevel=1) as fo:
    for a in range(0, n, 200_000):
        fo.write(b"".join(b"@r%d\n" % i + blk[i * 150:(i + 1) * 150] + b"\n+\n" + b"I" * 150 + b"\n" for i in range(a, min(n, a + 200_000))))
for thr in sys.argv[2:] or ["6", "8", "12", "16"]:
    os.environ["GS_GZ_THREADS"] = thr
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        _, _, tot = host.match_files(store, [gz])
        best = min(best, time.perf_counter() - t0)
    print(f"threads {thr}: {best:.2f} s -> {n * 150 / best / 1e9:.2f} Gbp/s", flush=True)
shutil.rmtree(d)
