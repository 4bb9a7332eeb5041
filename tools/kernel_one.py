"""a few full-size launches of ONE kernel for profiling (developer tool; tools/pmc_passes.sh runs it under rocprofv3):
    python tools/kernel_one.py [bench|miss|large|huge|filter]
bench: gs_match_kernel on the config-2 workload; miss: the same store, reads from genomes that are not in it;
large: gs_match_kernel on the 47 M-k-mer / 526-value store; huge: the 473 M-k-mer / 5 251-value store built on the device
(context-keyed gate); filter: gs_filter_kernel on the XOR index filter of the 47 M store"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "bench"
n = 10_000_000
if what == "huge":
    db = synth.SynthDB(k=31, genera=250, species_per_genus=20, build=False)
else:
    db = synth.SynthDB(genera=25, species_per_genus=20) if what in ("large", "filter") else synth.SynthDB()
src = synth.SynthDB(seed=43) if what == "miss" else db
gen = torch.from_numpy(src.genomes).cuda()
dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
synth.reads_device(gen, src.genomes.shape[0], src.genomes.shape[1], n, dseq, doff)
if what == "filter":
    keys = db.kmers[np.isin(db.value_idx, db.species_vi)]
    bits, hashes, factors = synth.xor_bloom_geometry(len(keys), 1e-8)
    dwords = torch.zeros((bits + 63) // 64, dtype=torch.int64, device="cuda")
    synth.xor_bloom_device(torch.from_numpy(keys).cuda(), len(keys), bits, torch.from_numpy(factors).cuda(), hashes, dwords)
    bloom = ga.DeviceBloomFilter(ga.BLOOM_XOR, bits, factors, dwords.cpu().numpy().view(np.uint64))
    del dwords
    m = ga.FastqBloomFilter(31, bloom, 1, 0.2, profile=True)
    acc = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        m.submit(dseq, doff, acc, n_reads=n)
else:
    if what == "huge":
        goff = torch.arange(db.genomes.shape[0] + 1, dtype=torch.int64, device="cuda") * db.genomes.shape[1]
        b = ga.DeviceDbBuilder(31, db.n_values, db.parent_vi)
        b.add(gen.reshape(-1), goff, db.species_vi, update=False)
        b.add(gen.reshape(-1), goff, db.species_vi, update=True)
        b.finish()
        store = b.to_store()
        b.close()
    else:
        store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
    for _ in range(3):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
m.sync()
l, t = m.kernel_time()
print(f"{what}: {t / l:.3f} ms/launch")
