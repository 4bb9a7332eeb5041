"""DB construction rate (developer tool): FillDBGoal + DBGoal over synthetic genomes on the device (gs_dbbuild) and, on a
sample, through the CPU restatement.      python tools/build_db_rate.py [genera ...]      (20 species per genus, 100 kbp each)"""
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

for genera in [int(x) for x in sys.argv[1:]] or [25, 250]:
    db = synth.SynthDB(genera=genera, species_per_genus=20)
    g = db.genomes
    seq = np.ascontiguousarray(g).reshape(-1)
    off = (np.arange(g.shape[0] + 1) * g.shape[1]).astype(np.uint64)
    dseq = torch.from_numpy(seq).cuda()
    doff = torch.from_numpy(off.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    res = {}
    for where, a, o in (("device", dseq, doff), ("host", seq, off)):
        t0 = time.time()
        b = ga.DeviceDbBuilder(31, db.n_values, db.parent_vi)
        b.add(a, o, db.species_vi, update=False)
        b.add(a, o, db.species_vi, update=True)
        t1 = time.time()
        keys, vals = b.finish()
        t2 = time.time()
        b.close()
        res[where] = {"add_s": round(t1 - t0, 3), "finish_fetch_s": round(t2 - t1, 3), "total_s": round(t2 - t0, 3)}
        ok = bool(np.array_equal(keys, db.kmers) and np.array_equal(vals, db.value_idx))
    # CPU restatement on the first 40 genomes (single thread, like one reader thread of the reference)
    ns = min(40, g.shape[0])
    sseq, soff = seq[:ns * g.shape[1]], off[:ns + 1]
    t0 = time.time()
    ob = orc.DbBuild(31, db.n_values, db.parent_vi)
    ob.fill(sseq, soff, db.species_vi[:ns])
    ob.optimize()
    ob.update(sseq, soff, db.species_vi[:ns])
    ob.fetch()
    ob.close()
    dt = time.time() - t0
    bases = int(seq.size)
    print(json.dumps({"genomes": int(g.shape[0]), "bases": bases, "pairs": 2 * bases, "distinct_kmers": int(len(db.kmers)),
                      "equals_synth_store": ok, "gs_dbbuild": res,
                      "mbases_per_s_device_resident": round(bases / res["device"]["total_s"] / 1e6, 1),
                      "cpu_restatement_mbases_per_s_1_thread": round(ns * g.shape[1] / dt / 1e6, 2)}), flush=True)
    del dseq, doff
