#!/bin/bash
# kernel times of the single-member gzip decoder (developer tool; on the GPU box):
#   bash tools/gunzip_kernels.sh [reads] [level] [tag]   -> gpurun_out/gunzip_kernels/<tag>_kernel_stats.csv + a summary on stdout
set -e -o pipefail
READS=${1:-4000000}
LEVEL=${2:-1}
TAG=${3:-run}
OUT=gpurun_out/gunzip_kernels
mkdir -p $OUT /tmp/gzkeep
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/$TAG -o $TAG --output-format csv -- python3 tools/gunzip_device_rate.py $READS $LEVEL /tmp/gzkeep > $OUT/$TAG.log 2>&1
f=$(find $OUT/$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${TAG}_kernel_stats.csv
grep -E "segments|text" $OUT/$TAG.log | tail -3
python3 - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("gi_"):
        print("  %-28s calls %3s  total %8.2f ms  mean %8.3f ms" % (r["Name"].split("(")[0], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
