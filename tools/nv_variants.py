"""cost of the per-taxid counter path: same store size, different numbers of value indices (developer tool)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

n = 10_000_000
for genera, spg, glen in ((4, 5, 100_000), (8, 9, 25_000), (10, 9, 20_000), (11, 10, 18_000), (12, 10, 16_000), (10, 20, 10_000), (25, 20, 4_000), (50, 40, 1_000)):
    db = synth.SynthDB(genera=genera, species_per_genus=spg, genome_len=glen)
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    gen = torch.from_numpy(db.genomes).cuda()
    dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
    doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
    for _ in range(2):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
    m.sync()
    l0, t0 = m.kernel_time()
    for _ in range(4):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
    m.sync()
    l1, t1 = m.kernel_time()
    print(f"values {db.n_values:5d}  k-mers {db.n_entries:8d}  table {store.info.table_bytes >> 20} MiB  "
          f"{(t1 - t0) / (l1 - l0):7.3f} ms/launch", flush=True)
    m.close()
    store.close()
    del dseq, doff, gen
