"""end-to-end rate of gs_host_filter_files: FASTQ on disk -> accepted / rest FASTQ files (developer tool).
filter_file_rate.py [reads] [plain|ml|fasta]: four-line FASTQ, FASTQ with the sequence over two lines, FASTA wrapped at 60"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402
from oracle import gs_oracle as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
shape = sys.argv[2] if len(sys.argv) > 2 else "plain"
db = synth.SynthDB()
keys = db.kmers[np.isin(db.value_idx, db.species_vi)]
ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
ob.put_many(keys)
gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
seq, off = synth.reads_host(db.genomes, n)
d = tempfile.mkdtemp(prefix="gsfrate")
path = os.path.join(d, "reads.fasta" if shape == "fasta" else "reads.fastq")
with open(path, "wb") as f:
    L = 150
    qual = b"I" * L
    for a in range(0, n, 200_000):
        b = min(n, a + 200_000)
        blk = seq[int(off[a]):int(off[b])].tobytes()
        if shape == "ml":
            f.write(b"".join(b"@r%d\n" % (a + i) + blk[i * L:i * L + 80] + b"\n" + blk[i * L + 80:(i + 1) * L] + b"\n+\n" + qual + b"\n"
                             for i in range(b - a)))
        elif shape == "fasta":
            f.write(b"".join(b">r%d\n" % (a + i) + blk[i * L:i * L + 60] + b"\n" + blk[i * L + 60:i * L + 120] + b"\n" +
                             blk[i * L + 120:(i + 1) * L] + b"\n" for i in range(b - a)))
        else:
            f.write(b"".join(b"@r%d\n" % (a + i) + blk[i * L:(i + 1) * L] + b"\n+\n" + qual + b"\n" for i in range(b - a)))
for fast in ("1", "0"):
    os.environ["GS_HOST_FAST"] = fast
    os.environ["GS_HOST_ML"] = fast  # ("0": the reference-exact parser thread for every shape)
    for out in (None, os.path.join(d, "acc.fastq")):
        t0 = time.perf_counter()
        tot = host.filter_files(gb, 31, [path], filtered_path=out)
        dt = time.perf_counter() - t0
        print(f"GS_HOST_FAST={fast} output={'file' if out else 'none'}: {dt:.2f} s -> {n * 150 / dt / 1e9:.2f} Gbp/s "
              f"({os.path.getsize(path) / dt / 1e9:.2f} GB/s of file), accepted {tot.filtered_reads}", flush=True)
shutil.rmtree(d)
