"""PCIe-inclusive rate of the host-batch path (gs_match_submit with GS_MEM_HOST): reads in pageable host
memory, staged to HBM by the library.  Reported in DESIGN.md next to the HBM-resident number of bench.py."""
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
for with_per_read in (False, True):
    cv = np.empty(n, np.int32) if with_per_read else None
    fl = np.empty(n, np.uint8) if with_per_read else None
    t0 = time.perf_counter()
    for _ in range(3):
        m.submit(seq, off, 0, cv, fl, n_reads=n)
    dt = (time.perf_counter() - t0) / 3
    print(f"host batch of {n} reads, per-read outputs={with_per_read}: {dt*1e3:.1f} ms -> {n*150/dt/1e9:.2f} Gbp/s "
          f"({n*150/dt/1e9:.2f} GB/s of sequence over PCIe)")
