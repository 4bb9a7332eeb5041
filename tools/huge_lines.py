"""Fabric line requests per read of the match kernel on the 473 M-k-mer store, by read stream (developer tool): run under
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d <dir> -o run -- python3 tools/huge_lines.py [reads]
and read <dir>/**/run_counter_collection.csv with tools/huge_lines.py --parse <dir> : two launches per stream, in the order
bench (half of the reads from the store), hit (every read has hits), miss (reads of other genomes)."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if "--parse" in sys.argv:
    d = sys.argv[sys.argv.index("--parse") + 1]
    n = int(sys.argv[sys.argv.index("--parse") + 2]) if len(sys.argv) > sys.argv.index("--parse") + 2 else 4_000_000
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "gs_match_kernel" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        by[int(r["Dispatch_Id"])]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    names = ["flags (2n reads)", "bench", "bench", "hit", "hit", "miss", "miss"]  # (the first launch picks the reads with hits)
    for (did, c), name in zip(sorted(by.items()), names):
        print(f"{name:16s} dispatch {did}: {c['ns'] / 1e6:7.3f} ms  read requests {c.get('TCC_EA0_RDREQ_sum', 0) / n:6.2f} / read  write requests "
              f"{c.get('TCC_EA0_WRREQ_sum', 0) / n:6.2f} / read")
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda", 0)
store, gen, g = bench._huge_store(ga, synth, torch, dev)[:3]
m = ga.FastqKMerMatcher(store)
dseq = torch.empty(2 * n * 150, dtype=torch.uint8, device=dev)
doff = torch.empty(2 * n + 1, dtype=torch.int64, device=dev)
synth.reads_device(gen, g.shape[0], g.shape[1], 2 * n, dseq, doff)
fl = torch.empty(2 * n, dtype=torch.uint8, device=dev)
m.submit(dseq, doff, 0, n_reads=2 * n, flags=fl)
m.sync()
hit = dseq.view(2 * n, 150)[(fl & 1) != 0][:n].contiguous().view(-1)
n_hit = hit.numel() // 150
other = synth.SynthDB(seed=43)
ogen = torch.from_numpy(other.genomes).to(dev)
miss = torch.empty(n * 150, dtype=torch.uint8, device=dev)
moff = torch.empty(n + 1, dtype=torch.int64, device=dev)
synth.reads_device(ogen, other.genomes.shape[0], other.genomes.shape[1], n, miss, moff)
torch.cuda.synchronize()
print("reads per stream", n, "hit stream", n_hit, flush=True)
for name, buf, cnt in (("bench", dseq, n), ("hit", hit, n_hit), ("miss", miss, n)):
    for _ in range(2):
        m.reset()
        m.submit_fixed(buf, 150, cnt)
    m.sync()
m.close()
store.close()
