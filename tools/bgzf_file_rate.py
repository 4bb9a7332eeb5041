"""file pipeline rate on a BGZF FASTQ for several inflating-thread counts (developer tool)"""
import os
import struct
import sys
import tempfile
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n)
blk = seq.tobytes()
text = b"".join(b"@r%d\n" % i + blk[i * 150:(i + 1) * 150] + b"\n+\n" + b"I" * 150 + b"\n" for i in range(n))
d = tempfile.mkdtemp(prefix="gsbgzf")
bz = os.path.join(d, "reads.fastq.gz")
with open(bz, "wb") as fo:
    for at in range(0, len(text) + 1, 65280):
        c = text[at:at + 65280]
        z = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = z.compress(c) + z.flush()
        fo.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body +
                 struct.pack("<II", zlib.crc32(c), len(c)))
for thr in sys.argv[2:] or ["4", "8", "12", "16"]:
    os.environ["GS_GZ_THREADS"] = thr
    for bg in ("1", "0"):
        os.environ["GS_BGZF"] = bg
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            _, _, tot = host.match_files(store, [bz])
            best = min(best, time.perf_counter() - t0)
        print(f"threads {thr} bgzf reader {bg}: {best:.2f} s -> {n * 150 / best / 1e9:.2f} Gbp/s ({len(text) / best / 1e9:.2f} GB/s of text)", flush=True)
