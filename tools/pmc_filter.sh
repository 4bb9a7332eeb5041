#!/bin/bash
# PMC passes over gs_filter_kernel (developer tool; run on the GPU box from the repo root)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/pmc_filter
mkdir -p "$out"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i + 1))
    timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d "$out/g$i" -o run -- python3 tools/filter_one.py > "$out/g$i.log" 2>&1 || echo "pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gs_filter_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{k},{sum(acc[k]) / len(acc[k]):.6g},{len(acc[k])}")
PY
