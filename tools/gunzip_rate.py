"""throughput of the gzip decoders of the ingest path on FASTQ text (developer tool): zlib, GsInflate, GsParallelGunzip"""
import gzip
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genestrip_amd import host, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
db = synth.SynthDB()
seq, off = synth.reads_host(db.genomes, n)
L = 150
blk = seq.tobytes()
rng = np.random.default_rng(1)
quals = [bytes(rng.choice(np.frombuffer(b"FFFFFFFFFFFFFFFF::,#", dtype=np.uint8), L).tolist()) for _ in range(256)]
text = b"".join(b"@A00123:45:HXX:1:1101:%d:%d 1:N:0:ACGT\n" % (i // 50, i * 37 % 40000) + blk[i * L:(i + 1) * L] + b"\n+\n" + quals[i & 255] + b"\n"
                for i in range(n))
comp = gzip.compress(text, compresslevel=6)
print(f"{len(text) / 1e9:.2f} GB text, ratio {len(text) / len(comp):.2f}", flush=True)
t0 = time.perf_counter(); zlib.decompress(comp, 31); dt = time.perf_counter() - t0
print(f"zlib {zlib.ZLIB_RUNTIME_VERSION}: {len(text) / dt / 1e9:.2f} GB/s", flush=True)
t0 = time.perf_counter(); host.gunzip(comp, len(text), 8 << 20); dt = time.perf_counter() - t0
print(f"GsInflate (1 thread, CRC checked inline): {len(text) / dt / 1e9:.2f} GB/s", flush=True)
for threads in (2, 4, 8, 12, 16):
    t0 = time.perf_counter(); host.gunzip_parallel(comp, len(text), threads, 1 << 20, 8 << 20); dt = time.perf_counter() - t0
    print(f"GsParallelGunzip x{threads} (+ CRC in the consumer): {len(text) / dt / 1e9:.2f} GB/s", flush=True)
