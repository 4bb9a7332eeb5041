"""Gbp/s of the match path by read length on the configs[1] store (developer tool): where the short-read kernel (<= 128 k-mer positions)
hands over to the long-read kernel (and, for batches of one length with 129 .. 192 positions, to the three-sub-round kernel).
    [GS_SWEEP_OFFSETS=1] python tools/read_len_sweep.py [len ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import genestrip_amd as ga
from genestrip_amd import synth

lens = [int(a) for a in sys.argv[1:]] or [100, 150, 158, 159, 170, 190, 222, 250, 286, 287, 350, 500]
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
gen = torch.from_numpy(db.genomes).cuda()
m = ga.FastqKMerMatcher(store)
for L in lens:
    n = 1_500_000_000 // L
    dseq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=L)
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        m.reset()
        m.sync()
        t0 = time.perf_counter()
        if os.environ.get("GS_SWEEP_OFFSETS"):
            m.submit(dseq, doff, 0, n_reads=n)  # the general call: an offsets array
        else:
            m.submit_fixed(dseq, L, n, 0)       # reads of one length, back to back
        m.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{L:5d} bp ({L - 30:4d} positions): {best * 1e3:7.2f} ms for {n} reads = {n * L / best / 1e9:6.1f} Gbp/s", flush=True)
    del dseq, doff
