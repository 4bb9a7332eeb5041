"""Store-size sweep of the fused match kernel on ONE GPU (developer tool): BASELINE.json configs[4] asks for a 500 M-k-mer
store split over 8 GPUs; with 288 GB of HBM the whole store fits one device (records + table + gate ~ 10 GB), so the
question is what the kernel does there.      python tools/bench_scale.py [genera ...]      (20 species per genus, 100 kbp each)
Prints one JSON line per size: build seconds, device bytes, kernel ms per 10 M reads, oracle spot check."""
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

n = 10_000_000
for genera in [int(x) for x in sys.argv[1:]] or [25, 100, 250]:
    t0 = time.time()
    db = synth.SynthDB(genera=genera, species_per_genus=20)
    t_db = time.time() - t0
    t0 = time.time()
    store = ga.DeviceKMerStore(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    t_store = time.time() - t0
    info = store.info
    gen = torch.from_numpy(db.genomes).cuda()
    dseq = torch.empty(n * 150, dtype=torch.uint8, device="cuda")
    doff = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff)
    m = ga.FastqKMerMatcher(store, ga.MatchConfig(profile=True))
    for _ in range(2):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
    m.sync()
    l0, ms0 = m.kernel_time()
    for _ in range(4):
        m.reset()
        m.submit(dseq, doff, 0, n_reads=n)
    m.sync()
    l1, ms1 = m.kernel_time()
    kms = (ms1 - ms0) / (l1 - l0)
    nchk = 50_000
    seq, off = synth.reads_host(db.genomes, nchk)
    odb = orc.DB(31, db.kmers, db.value_idx, db.n_values, db.parent_vi)
    orun = orc.MatchRun(odb)
    orun.submit(seq, off, threads=16, per_read=False)
    ot, _ = orun.finish()
    odb.close()
    m.reset()
    m.submit(dseq, doff, 0, n_reads=nchk)
    gt, _ = m.finish()
    print(json.dumps({"store_kmers": int(db.n_entries), "n_values": int(db.n_values), "in_records": int(info.n_in_records),
                      "rec_bytes": int(info.rec_bytes), "table_bytes": int(info.table_bytes), "mgate_bytes": int(info.mgate_bytes),
                      "build_s": {"synthetic_arrays": round(t_db, 1), "gs_db_create": round(t_store, 1)},
                      "kernel_ms": round(kms, 3), "gbps": round(n * 150 / kms / 1e6, 1),
                      "bit_exact_50k": bool(np.array_equal(ot, gt))}), flush=True)
    m.close()
    store.close()
    del gen, dseq, doff, db
    torch.cuda.empty_cache()
