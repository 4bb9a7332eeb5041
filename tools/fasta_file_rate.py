"""FASTA contigs through the file pipeline (developer tool): records found and gathered on the device (gs_match_submit_fasta; contigs
longer than 158 bases take the long-read kernel) against the reference-exact parser on one thread (GS_HOST_FAST=0).
python tools/fasta_file_rate.py [Mbases] [contig_len]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import host, synth  # noqa: E402

total = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 600_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1200  # a multiple of 60: every line is full
n = total // L
db = synth.SynthDB(k=31)
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n, read_len=L)
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gs_fa_%d.fasta" % os.getpid())
lines = seq.reshape(n, L // 60, 60)
body = np.concatenate([lines, np.full((n, L // 60, 1), 10, dtype=np.uint8)], axis=2).reshape(n, -1)
hdr = np.frombuffer(b">contig length=%7d\n" % L, dtype=np.uint8)
np.concatenate([np.broadcast_to(hdr, (n, len(hdr))), body], axis=1).tofile(path)
size = os.path.getsize(path)
print("wrote %.2f GB (%d contigs of %d bases, wrapped at 60)" % (size / 1e9, n, L), flush=True)
for fast in ("1", "0", "1"):
    os.environ["GS_HOST_FAST"] = fast
    t0 = time.time()
    table, _, tot = host.match_files(store, [path])
    dt = time.time() - t0
    print("GS_HOST_FAST=%s  %.2f s  %.2f Gbp/s  %.2f GB/s of file  reads %d  classified %d" %
          (fast, dt, n * L / dt / 1e9, size / dt / 1e9, tot.reads, int(table[:, 0].sum())), flush=True)
os.unlink(path)
