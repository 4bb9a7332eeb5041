"""rate of the device DEFLATE decoder on BGZF FASTQ (developer tool): python tools/inflate_rate.py [reads] [level]
wall time of gs_inflate_members (H2D of the compressed bytes + kernel + D2H of the text); run under
rocprofv3 --kernel-trace --stats for the kernel's own time"""
from concurrent.futures import ThreadPoolExecutor
import os
import struct
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402
import bench  # noqa: E402


def _blk(args):
    c, level = args
    z = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = z.compress(c) + z.flush()
    return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(c), len(c))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    db = synth.SynthDB()
    seq, off = synth.reads_host(db.genomes, n)
    text = bench._fastq_text(seq, n).tobytes()
    with ThreadPoolExecutor(16) as pool:  # (threads: zlib releases the interpreter lock; nothing here may fork once HIP is loaded)
        parts = list(pool.map(_blk, [(text[a:a + 65280], level) for a in range(0, len(text), 65280)], chunksize=64))
    data = b"".join(parts)
    members, reached = ga.bgzf_members(data)
    assert reached == len(data)
    print(f"{len(text) / 1e6:.0f} MB text, {len(data) / 1e6:.0f} MB BGZF level {level}, {len(members)} members", flush=True)
    for _ in range(3):
        t0 = time.perf_counter()
        out, st = ga.inflate_members(data, members)
        dt = time.perf_counter() - t0
        print(f"gs_inflate_members: {dt * 1e3:.1f} ms wall = {len(text) / dt / 1e9:.2f} GB/s of text incl. copies", flush=True)
    assert out.tobytes() == text
    print("text equals the input")
