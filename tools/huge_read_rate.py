"""match rate of records of 30 kbp .. 5 Mbp (contigs, chromosomes in a FASTA) -- developer tool.  Such a record is cut into chunks over
many waves (gs_match_huge_kernel); GS_HUGE_MIN=2000000000 in the environment keeps it on one wave (the long-read path) for comparison.
    python tools/huge_read_rate.py [case ...]      case = records x length, e.g. 1x5000000 8x5000000 256x100000"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(1, 5_000_000), (8, 5_000_000), (64, 1_000_000), (256, 100_000), (256, 40_000)]
K = 31
shape = [int(x) for x in os.environ.get("GS_RATE_SHAPE", "4,5,100000").split(",")]
db = synth.SynthDB(k=K, genera=shape[0], species_per_genus=shape[1], genome_len=shape[2])
print("store: %d k-mers, %d values; GS_HUGE_MIN=%s GS_HUGE_CHUNK=%s" % (db.n_entries, db.n_values, os.environ.get("GS_HUGE_MIN", "-"), os.environ.get("GS_HUGE_CHUNK", "-")),
      flush=True)
store = ga.DeviceKMerStore(K, db.kmers, db.value_idx, db.n_values, db.parent_vi)
m = ga.FastqKMerMatcher(store)
rng = np.random.default_rng(1)
g0 = db.genomes
for n, L in cases:
    recs = []
    for _ in range(n):  # a record: pieces of 1 .. 60 kbp of the genomes, as an assembly of a mixed sample would hold them
        parts, left = [], L
        while left > 0:
            s = int(rng.integers(0, g0.shape[0]))
            c = min(left, int(rng.integers(1000, 60000)))
            p = int(rng.integers(0, g0.shape[1] - c))
            parts.append(g0[s][p:p + c])
            left -= c
        recs.append(np.concatenate(parts))
    seq = np.concatenate(recs)
    off = np.arange(n + 1, dtype=np.int64) * L
    dseq, doff = torch.from_numpy(seq).cuda(), torch.from_numpy(off).cuda()
    best = 1e9
    for _ in range(4):
        m.reset()
        m.sync()
        t0 = time.perf_counter()
        m.submit(dseq, doff, 0, n_reads=n)
        m.sync()
        best = min(best, time.perf_counter() - t0)
    table, _ = m.finish()
    print(f"{n:4d} records of {L:8d} bp: {best * 1e3:9.3f} ms -> {n * L / best / 1e9:8.2f} Gbp/s   (k-mers found: {int(table[:, 2].sum())})", flush=True)
    m.reset()
    t0 = time.perf_counter()
    seg_off = m.segments(seq, off)[0]  # (stages the batch: the copy is part of the time)
    dt = time.perf_counter() - t0
    print(f"     Kraken-style segments (from host memory, two passes, fetched): {int(seg_off[-1])} runs, {dt * 1e3:9.3f} ms", flush=True)
    del dseq, doff
