"""single-member gzip streams through the device decoder, with the first mismatch against the input (developer tool; run from the repo
root on the GPU box, under `timeout`)"""
import os, sys, time, zlib, gzip
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import genestrip_amd as ga

T0 = time.time()


def run(name, data, level=6, **kw):
    z = gzip.compress(data, compresslevel=level, mtime=0)
    print("%.1f s case %s: %d -> %d" % (time.time() - T0, name, len(data), len(z)), flush=True)
    try:
        got, info = ga.gunzip_device(z, len(data))
    except ga.GsError as e:
        print("   raised", e.code, str(e)[:200], flush=True)
        return
    g = got.tobytes()
    first = next((i for i in range(min(len(g), len(data))) if g[i] != data[i]), None)
    print("   info", info, "equal", g == data, "first mismatch", first, flush=True)
    if first is not None:
        print("   want", data[max(0, first - 8):first + 24], "\n   got ", g[max(0, first - 8):first + 24], flush=True)


rng = np.random.default_rng(5)
acgt = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 3_000_000))
run("tiny", b"hello hello hello\n")
run("acgt 100k", acgt[:100_000])
run("acgt 3M", acgt)
rec = b"".join(b"@r%07d\n" % i + acgt[i * 150:(i + 1) * 150] + b"\n+\n" + b"I" * 150 + b"\n" for i in range(15000))
run("fastq 4.7M level 6", rec)
run("fastq 4.7M level 1", rec, level=1)
run("fastq 4.7M level 9", rec, level=9)
