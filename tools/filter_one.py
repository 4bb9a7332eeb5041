"""a few launches of gs_filter_kernel on a config-3-scale XOR index filter, for profiling (developer tool)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402
from oracle import gs_oracle as orc  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
db = synth.SynthDB(genera=25, species_per_genus=20)
keys = db.kmers[np.isin(db.value_idx, db.species_vi)]
ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
ob.put_many(keys)
dbloom = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
gen = torch.from_numpy(db.genomes).cuda()
dseq = torch.empty(nf * 150, dtype=torch.uint8, device="cuda")
doff = torch.empty(nf + 1, dtype=torch.int64, device="cuda")
synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], nf, dseq, doff)
flt = ga.FastqBloomFilter(31, dbloom, 1, 0.2, profile=True)
acc = torch.empty(nf, dtype=torch.uint8, device="cuda")
for _ in range(3):
    flt.submit(dseq, doff, acc, n_reads=nf)
flt.sync()
l, t = flt.kernel_time()
print(f"{t / l:.3f} ms/launch, accepted {float(acc.float().mean()):.4f}")
