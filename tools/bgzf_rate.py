"""decode rate of the host gzip readers on BGZF and ordinary gzip input (developer tool; CPU only)"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from genestrip_amd import host
from conftest import bgzf
import zlib, gzip
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
recs = []
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
seq = acgt[rng.integers(0, 4, n * 150)].tobytes()
data = b"".join(b"@r%d\n" % i + seq[i*150:(i+1)*150] + b"\n+\n" + b"I" * 150 + b"\n" for i in range(n))
raw = bgzf(data, level=1)
gzr = gzip.compress(data, 1)
print(len(data) / 1e6, "MB text", len(raw) / 1e6, "MB bgzf", len(gzr)/1e6, "MB gz")
for label, blob in (("bgzf", raw), ("gzip", gzr)):
    for thr in (1, 4, 8, 16):
        t0 = time.perf_counter()
        for _ in range(3):
            out = host.gunzip_parallel(blob, len(data), thr, 1 << 20, 8 << 20)
        dt = (time.perf_counter() - t0) / 3
        assert out == data
        print(label, thr, "threads", round(len(data) / dt / 1e9, 2), "GB/s")
t0 = time.perf_counter(); zlib.decompress(gzr, 31); print("zlib", round(len(data) / (time.perf_counter() - t0) / 1e9, 2), "GB/s")
