#!/bin/bash
# instruction counters of gs_match_kernel per build variant and read stream (developer tool, GPU box):
#   tools/pmc_variants.sh "a0 a1 a2 a3" "hit miss"
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for v in $1; do
  for st in $2; do
    out=gpurun_out/pmcv_${v}_${st}
    rm -rf "$out"; mkdir -p "$out"
    GS_LIBGSGPU=$PWD/build/ablate/libgsgpu_$v.so GS_STREAM_CHILD=1 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
        --output-format csv -d "$out" -o run -- python3 tools/stream_times.py --stream=$st > "$out.log" 2>&1 || echo "failed $v $st"
    python3 - "$out" "$v" "$st" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gs_match_kernel" in r["Kernel_Name"]]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows}
    longest = max(dur.values()) if dur else 0
    for r in rows:
        if dur[r["Dispatch_Id"]] >= 0.8 * longest:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], sys.argv[3], " ".join(f"{k}={sum(v)/len(v)/1e7:.1f}/read" for k, v in sorted(acc.items())))
PY
  done
done
