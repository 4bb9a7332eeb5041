"""per-kernel summary of the rocprofv3 --pmc passes written by tools/pmc_passes.sh: mean counter value over the
FULL-SIZE launches of one kernel (argv[2], default gs_match_kernel) (launches at least 80 % as long as the longest one of their pass)"""
import collections
import csv
import glob
import sys

out = sys.argv[1].rstrip("/")
kernel = sys.argv[2] if len(sys.argv) > 2 else "gs_match_kernel"
acc = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    if not rows:
        continue
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows}
    longest = max(dur.values())
    for r in rows:
        if dur[r["Dispatch_Id"]] >= 0.8 * longest:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + ".csv", "w") as o:
    o.write("counter,value_per_launch,launches\n")
    for name in sorted(acc):
        o.write(f"{name},{sum(acc[name]) / len(acc[name]):.6g},{len(acc[name])}\n")
print(open(out + ".csv").read())
