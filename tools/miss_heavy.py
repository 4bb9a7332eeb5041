"""match kernel time on a miss-only read stream (reads drawn from genomes that are NOT in the store): the common
case of targeted stores screened against host/background DNA (developer tool)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

n = 10_000_000
db = synth.SynthDB(seed=42)
other = synth.SynthDB(seed=43)
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
gen = torch.from_numpy(other.genomes).cuda()
dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
synth.reads_device(gen, other.genomes.shape[0], other.genomes.shape[1], n, dseq, doff)
m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
for _ in range(2):
    m.reset()
    m.submit(dseq, doff, 0, n_reads=n)
t, _ = m.finish()
l0, t0 = m.kernel_time()
for _ in range(5):
    m.reset()
    m.submit(dseq, doff, 0, n_reads=n)
m.sync()
l1, t1 = m.kernel_time()
ms = (t1 - t0) / (l1 - l0)
print(f"miss-only stream: {ms:.3f} ms/launch -> {n * 150 / ms / 1e6:.1f} Gbp/s  (hits in table: {int(t[:, 2].sum())}, "
      f"gate {store.info.gate_bytes} B, mgate {store.info.mgate_bytes} B)")
