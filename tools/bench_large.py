"""Large-store measurements (developer tool; BASELINE.json configs 3/4 per GPU): a ~50 M-k-mer store that no longer
fits the Infinity Cache -- match kernel time without the L2 gate, filter kernel time on the XOR index filter, both
with an oracle spot check.

    python tools/bench_large.py [genera] [species_per_genus] [reads] [filter_reads]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402
from oracle import gs_oracle as orc  # noqa: E402

genera = int(sys.argv[1]) if len(sys.argv) > 1 else 25
spg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
nf = int(sys.argv[4]) if len(sys.argv) > 4 else 20_000_000

t0 = time.time()
db = synth.SynthDB(genera=genera, species_per_genus=spg)
print(f"store: {db.n_entries} k-mers, {db.n_values} values, built in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
info = store.info
print(f"gs_db_create: {time.time() - t0:.1f} s, table {info.table_bytes / 2**20:.0f} MiB, gate {info.gate_bytes} B, "
      f"max displacement {info.max_displacement}", flush=True)

dev = torch.device("cuda")
gen = torch.from_numpy(db.genomes).to(dev)
nmax = max(n, nf)
dseq = torch.empty(nmax * 150, dtype=torch.uint8, device=dev)
doff = torch.empty(nmax + 1, dtype=torch.int64, device=dev)
synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], nmax, dseq, doff)
out = {}

# ---- match
m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
for _ in range(2):
    m.reset()
    m.submit(dseq, doff, 0, n_reads=n)
m.sync()
l0, ms0 = m.kernel_time()
for _ in range(5):
    m.reset()
    m.submit(dseq, doff, 0, n_reads=n)
m.sync()
l1, ms1 = m.kernel_time()
kms = (ms1 - ms0) / max(1, l1 - l0)
out["match"] = {"reads": n, "kernel_ms": round(kms, 3), "gbp_s": round(n * 150 / kms / 1e6, 2),
                "frac_of_8TBs": round(n * 7830 / (kms * 1e-3) / 8e12, 4)}
print("match:", out["match"], flush=True)
nchk = 100_000
seq, off = synth.reads_host(db.genomes, nchk)
odb = orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
orun = orc.MatchRun(odb)
orun.submit(seq, off, threads=16, per_read=False)
ot, _ = orun.finish()
m.reset()
m.submit(dseq, doff, 0, n_reads=nchk)
gt, _ = m.finish()
out["match"]["bit_exact_100k"] = bool(np.array_equal(ot, gt))
print("match parity:", out["match"]["bit_exact_100k"], flush=True)
m.close()

# ---- filter (index filter over the species k-mers = requested taxa)
if nf <= 0:  # match only (the PMC passes of the match kernel use filter_reads = 0)
    print(json.dumps(out))
    sys.exit(0)
keys = db.kmers[np.isin(db.value_idx, db.species_vi)]
t0 = time.time()
ob = orc.Bloom(orc.BLOOM_XOR, len(keys), 1e-8)
ob.put_many(keys)
print(f"XOR Bloom: {len(keys)} keys, {ob.bits / 8 / 2**20:.0f} MiB, {ob.hashes} hashes, built in {time.time() - t0:.1f} s", flush=True)
gb = ga.DeviceBloomFilter(ga.BLOOM_XOR, ob.bits, ob.hash_factors, ob.words)
flt = ga.FastqBloomFilter(31, gb, 1, 0.2, profile=True)
acc = torch.empty(nf, dtype=torch.uint8, device=dev)
for _ in range(2):
    flt.submit(dseq, doff, acc, n_reads=nf)
flt.sync()
l0, ms0 = flt.kernel_time()
for _ in range(5):
    flt.submit(dseq, doff, acc, n_reads=nf)
flt.sync()
l1, ms1 = flt.kernel_time()
kms = (ms1 - ms0) / max(1, l1 - l0)
want = ob.filter_batch(31, 1, 0.2, seq, off, threads=16)
got = acc[:nchk].cpu().numpy()
out["filter"] = {"reads": nf, "kernel_ms": round(kms, 3), "gbp_s": round(nf * 150 / kms / 1e6, 2),
                 "frac_of_8TBs": round(nf * 7830 / (kms * 1e-3) / 8e12, 4), "accepted_frac": round(float(acc.float().mean()), 4),
                 "bit_exact_100k": bool(np.array_equal(want, got))}
print("filter:", out["filter"], flush=True)
print(json.dumps(out))
