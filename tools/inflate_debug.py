"""device DEFLATE on a handful of single members, with the first mismatch against the input (developer tool; run from the repo root on the GPU box,
under `timeout`: a decoder bug can be a kernel that does not end)"""
import os, sys, zlib, struct, gzip
sys.path.insert(0, os.getcwd())  # run from the repo root
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import genestrip_amd as ga
from conftest import GOLDEN, bgzf

def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    z = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = z.compress(data) + z.flush()
    bsize = 18 + len(body) + 8 - 1
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + body + struct.pack("<II", zlib.crc32(data), len(data)))

import time
T0 = time.time()
def run(name, data, **kw):
    f = member(data, **kw)
    print("%.1f s case" % (time.time() - T0), name, len(data), "->", len(f), flush=True)
    members, reached = ga.bgzf_members(f)
    got, st = ga.inflate_members(f, members)
    g = got.tobytes()
    ok = g == data and not st.any()
    first = next((i for i in range(min(len(g), len(data))) if g[i] != data[i]), None)
    print("   status", st.tolist(), "equal", g == data, "first mismatch", first, flush=True)
    if first is not None:
        print("   want", data[max(0, first - 8):first + 24], "\n   got ", g[max(0, first - 8):first + 24], flush=True)
    return ok

rng = np.random.default_rng(5)
acgt = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 65000))
run("x fixed", b"x", strategy=zlib.Z_FIXED)
run("hello fixed", b"hello hello hello hello", strategy=zlib.Z_FIXED)
run("abc dyn", b"abcdefghijklmnopqrstuvwxyz" * 40)
run("acgt 300", acgt[:300])
run("acgt 3000 fixed", acgt[:3000], strategy=zlib.Z_FIXED)
run("acgt 65000", acgt)
run("runs", (b"A" * 300 + b"CG" * 200 + b"\n") * 80, level=9)
text = gzip.open(os.path.join(GOLDEN, "human_virus", "sample.fastq.gz")).read()
run("sample 60000", text[:60000])
