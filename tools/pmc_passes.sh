#!/bin/bash
# PMC passes over one kernel (developer tool; run on the GPU box from the repo root):
#   [GS_PMC_KERNEL=gs_filter_kernel] tools/pmc_passes.sh <tag> [python script + args ...]
# One rocprofv3 --pmc run per counter group (never combined with trace domains), then a per-kernel summary
# (mean over the full-size launches) in gpurun_out/pmc_<tag>.csv.
set -u
tag=${1:-x}
shift
if [ $# -eq 0 ]; then set -- tools/kernel_one.py; fi
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p "$out"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d "$out/g$i" -o run -- python3 "$@" > "$out/g$i.log" 2>&1 || echo "pass $i failed" >> "$out/fail.log"
    echo "pass $i done: $grp"
done
python3 tools/pmc_summary.py "$out" "${GS_PMC_KERNEL:-gs_match_kernel}"
