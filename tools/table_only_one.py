"""the k = 16 bucket-table workload of bench.py's `table_only` object alone (developer tool; for rocprofv3 --pmc passes, tools/pmc_passes.sh):
python tools/table_only_one.py [reads] [k]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import genestrip_amd as ga  # noqa: E402
from genestrip_amd import synth  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    from oracle import gs_oracle as orc
    print(bench.leg_table_only(ga, synth, orc, torch, torch.device("cuda", 0), 8, k=k, n=n, nchk=20_000))
