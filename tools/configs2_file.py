"""BASELINE.json configs[2] end to end as bench.py's `filter.configs2_file` leg runs it, at any read count (developer tool):
configs2_file.py [reads] [--no-plain]: BGZF FASTQ written on the device -> gs_host_filter_files -> accepted reads as .gz."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16_000_000
    dev = torch.device("cuda", 0)
    db = synth.SynthDB(k=bench.K, genera=25, species_per_genus=20)
    gen = torch.from_numpy(db.genomes).to(dev)
    dseq = torch.empty(n * bench.READ_LEN, dtype=torch.uint8, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    synth.reads_device(gen, db.genomes.shape[0], db.genomes.shape[1], n, dseq, doff, read_len=bench.READ_LEN)
    bloom, _ = bench._index_filter(ga, synth, torch, dev, db)
    flt = ga.FastqBloomFilter(bench.K, bloom, 1, 0.2)
    acc = torch.empty(n, dtype=torch.uint8, device=dev)
    flt.submit(dseq, doff, acc, n_reads=n)
    flt.sync()
    res = bench.leg_configs2_file(ga, torch, bloom, dseq, n, acc, bench._usable_cores(), check_plain="--no-plain" not in sys.argv)
    print(json.dumps(res, indent=1))
