"""kernel-time ablations of gs_match_kernel on the config-2 workload (developer tool)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
gen = torch.from_numpy(db.genomes).cuda()
dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff)
for name, cfg in (("default", {}), ("no classify", dict(classify=False)), ("no unique", dict(count_unique=False)),
                  ("no classify, no unique", dict(classify=False, count_unique=False))):
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True, **cfg))
    for _ in range(2):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
    m.sync()
    l0, t0 = m.kernel_time()
    for _ in range(5):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
    m.sync()
    l1, t1 = m.kernel_time()
    print(f"{name:26s} {(t1 - t0) / (l1 - l0):8.3f} ms/launch  -> {n * 150 / ((t1 - t0) / (l1 - l0) * 1e-3) / 1e9:6.1f} Gbp/s")
    m.close()
