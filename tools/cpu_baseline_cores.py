"""CPU restatement (oracle) rate on 1 thread and on all usable threads, same read stream as bench.py (developer tool)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genestrip_amd import synth  # noqa: E402
from oracle import gs_oracle as orc  # noqa: E402

db = synth.SynthDB()
odb = orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi, bloom_gate=True)
for threads, n in ((1, 400_000), (len(os.sched_getaffinity(0)), 4_000_000)):
    seq, off = synth.reads_host(db.genomes, n)
    run = orc.MatchRun(odb)
    t0 = time.perf_counter()
    run.submit(seq, off, threads=threads, per_read=False)
    dt = time.perf_counter() - t0
    print(f"{threads:3d} thread(s): {n} reads in {dt:.2f} s -> {n * 150 / dt / 1e9:.4f} Gbp/s")
