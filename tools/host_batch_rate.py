"""PCIe-inclusive rate of the host-batch path (gs_match_submit with GS_MEM_HOST): reads in pageable and in page-locked
host memory, staged to HBM by the library.  Reported in DESIGN.md next to the HBM-resident number of bench.py."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
db = synth.SynthDB()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
seq, off = synth.reads_host(db.genomes, n)
m = ga.FastqKMerMatcher(store)
m.submit(seq, off, 0, n_reads=n)  # warm-up (allocates the staging buffers)
m.reset()
import torch  # noqa: E402  (page-locked arrays: what gs_pinned_alloc gives a C or Java host)

pseq = torch.from_numpy(seq).pin_memory().numpy()
poff = torch.from_numpy(off.astype(np.int64)).pin_memory().numpy().view(off.dtype)
for label, s_, o_ in (("pageable", seq, off), ("page-locked", pseq, poff)):
    for with_per_read in (False, True):
        if with_per_read:
            cv = torch.empty(n, dtype=torch.int32).pin_memory().numpy() if label == "page-locked" else np.empty(n, np.int32)
            fl = torch.empty(n, dtype=torch.uint8).pin_memory().numpy() if label == "page-locked" else np.empty(n, np.uint8)
        else:
            cv = fl = None
        t0 = time.perf_counter()
        for _ in range(3):
            m.submit(s_, o_, 0, cv, fl, n_reads=n)
        m.sync()
        dt = (time.perf_counter() - t0) / 3
        print(f"{label} host batch of {n} reads, per-read outputs={with_per_read}: {dt*1e3:.1f} ms -> {n*150/dt/1e9:.2f} Gbp/s "
              f"({n*150/dt/1e9:.2f} GB/s of sequence over PCIe)")

# asynchronous batches from page-locked arrays: the copy of batch i+1 under the kernel of batch i (gs_match_submit_async)
half = n // 2
parts = []
for a, b in ((0, half), (half, n)):
    ps = torch.from_numpy(seq[int(off[a]):int(off[b])].copy()).pin_memory().numpy()
    po = torch.from_numpy((off[a:b + 1] - off[a]).astype(np.int64)).pin_memory().numpy().view(off.dtype)
    parts.append((ps, po, a))
m.reset()
t0 = time.perf_counter()
rounds = 6
last = []
for r in range(rounds):
    for ps, po, a in parts:
        last.append(m.submit_async(ps, po, a))
        if len(last) > 2:
            m.wait(last[-3])
m.sync()
dt = (time.perf_counter() - t0) / rounds
print(f"page-locked host batches of {half} reads, asynchronous (two under way): {dt*1e3:.1f} ms per {n} reads -> {n*150/dt/1e9:.2f} Gbp/s")
