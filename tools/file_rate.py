"""end-to-end rate of the file pipeline (gs_host_match_files): FASTQ on disk -> parse -> stage -> match -> table
(developer tool).  argv[1] = number of reads (default 8 M), argv[2] = 'gz' to also time a gzip-compressed copy."""
import gzip
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n)
d = tempfile.mkdtemp(prefix="gsrate")
path = os.path.join(d, "reads.fastq")
t0 = time.perf_counter()
with open(path, "wb") as f:
    L = 150
    qual = b"I" * L
    chunk = 200_000
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        parts = []
        blk = seq[int(off[a]):int(off[b])].tobytes()
        for i in range(b - a):
            parts.append(b"@r%d\n" % (a + i))
            parts.append(blk[i * L:(i + 1) * L])
            parts.append(b"\n+\n")
            parts.append(qual)
            parts.append(b"\n")
        f.write(b"".join(parts))
size = os.path.getsize(path)
print(f"wrote {path}: {size / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s", flush=True)


def run(p, label):
    t0 = time.perf_counter()
    res = host.match_files(store, [p])
    dt = time.perf_counter() - t0
    tot = (res[2].reads, res[2].kmers, res[2].bps) if hasattr(res[2], "reads") else res[2]
    print(f"{label}: {dt:.2f} s -> {n * 150 / dt / 1e9:.2f} Gbp/s ({os.path.getsize(p) / dt / 1e9:.2f} GB/s of file)  totals={tot}",
          flush=True)


run(path, "plain FASTQ (cold-ish)")
run(path, "plain FASTQ (page cache)")
if len(sys.argv) > 2 and sys.argv[2] == "gz":
    gz = path + ".gz"
    t0 = time.perf_counter()
    with open(path, "rb") as fi, gzip.open(gz, "wb", compresslevel=1) as fo:
        shutil.copyfileobj(fi, fo, 1 << 24)
    print(f"gzip -1: {time.perf_counter() - t0:.1f} s, {os.path.getsize(gz) / 1e9:.2f} GB", flush=True)
    run(gz, "gzip FASTQ")
if len(sys.argv) > 2 and sys.argv[2] == "bgzf":  # what bgzip writes: members of <= 64 KiB that say how long they are
    import struct
    import zlib
    bz = path + ".bgzf.gz"
    t0 = time.perf_counter()
    with open(path, "rb") as fi, open(bz, "wb") as fo:
        while True:
            c = fi.read(65280)
            z = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = z.compress(c) + z.flush()
            fo.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body +
                     struct.pack("<II", zlib.crc32(c), len(c)))
            if not c:
                break
    print(f"bgzf -1: {time.perf_counter() - t0:.1f} s, {os.path.getsize(bz) / 1e9:.2f} GB", flush=True)
    run(bz, "BGZF FASTQ")
    run(bz, "BGZF FASTQ (again)")
shutil.rmtree(d)
