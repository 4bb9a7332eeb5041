"""a few full-size launches of gs_match_kernel on the config-2 workload, for profiling (developer tool);
argv[1] = 'miss' draws the reads from genomes that are not in the store"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

n = 10_000_000
db = synth.SynthDB()
src = synth.SynthDB(seed=43) if len(sys.argv) > 1 and sys.argv[1] == "miss" else db
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
gen = torch.from_numpy(src.genomes).cuda()
dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
synth.reads_device(gen, src.genomes.shape[0], src.genomes.shape[1], n, dseq, doff)
m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
for _ in range(3):
    m.reset()
    m.submit(dseq, doff, 0, n_reads=n)
m.sync()
l, t = m.kernel_time()
print(f"{t / l:.3f} ms/launch")
