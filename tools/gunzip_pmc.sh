#!/bin/bash
# instruction / wait counters of the single-member gzip decoder's two big kernels (developer tool; on the GPU box):
#   bash tools/gunzip_pmc.sh [reads] [level]   -> gpurun_out/pmc_gunzip/ + the per-launch means on stdout
# One rocprofv3 --pmc run per counter group (never combined with trace domains).
READS=${1:-4000000}
LEVEL=${2:-1}
export TMPDIR=/tmp
out=gpurun_out/pmc_gunzip
mkdir -p $out /tmp/gzkeep
python3 tools/gunzip_device_rate.py $READS $LEVEL /tmp/gzkeep > $out/base.log 2>&1 || exit 1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
    i=$((i + 1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$out/g$i" -o run -- python3 tools/gunzip_device_rate.py $READS $LEVEL /tmp/gzkeep > "$out/g$i.log" 2>&1
    rc=$?
    if [ $rc -ge 124 ]; then echo "pass $i killed"; exit 1; fi
done
echo "# gi_segment_kernel (per launch)"
python3 tools/pmc_summary.py "$out" gi_segment_kernel
echo "# gi_find_kernel (per launch; several launches per batch)"
python3 tools/pmc_summary.py "$out" gi_find_kernel
