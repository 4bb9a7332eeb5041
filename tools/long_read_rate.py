"""match rate by read length at a constant number of bases (developer tool): reads of 150 bp take gs_match_kernel, longer ones
gs_match_long_kernel (one wave per read, 128 k-mer positions per iteration).   python tools/long_read_rate.py [total_Mbases [lengths ...]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

total = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 1_500_000_000
lengths = [int(x) for x in sys.argv[2:]] or [150, 300, 1000, 5000, 20000, 90000]
K = int(os.environ.get("GS_RATE_K", "31"))  # GS_RATE_K=25: the kernels with k as a run-time value
shape = [int(x) for x in os.environ.get("GS_RATE_SHAPE", "4,5,100000").split(",")]  # genera, species per genus, genome length
db = synth.SynthDB(k=K, genera=shape[0], species_per_genus=shape[1], genome_len=shape[2])
print("store: %d k-mers, %d values" % (db.n_entries, db.n_values), flush=True)
gen = torch.from_numpy(db.genomes).cuda()
store = ga.DeviceKMerStore(K, db.kmers, db.value_idx, db.n_values, db.parent_vi)
m = ga.FastqKMerMatcher(store)
for L in lengths:
    n = total // L
    dseq = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=L)
    best = 1e9
    for _ in range(3):
        m.reset()
        m.sync()
        t0 = time.perf_counter()
        m.submit(dseq, doff, 0, n_reads=n)
        m.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"read length {L:6d}: {n:9d} reads, {best * 1e3:8.2f} ms -> {n * L / best / 1e9:7.1f} Gbp/s", flush=True)
    del dseq, doff
