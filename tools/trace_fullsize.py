#!/usr/bin/env python3
"""Per kernel of a rocprofv3 --kernel-trace CSV: all launches, and the FULL-SIZE launches (duration >= 0.8 x the longest one)
with their mean -- the number that must agree with the HIP-event kernel time of bench.py's JSON line (the plain --stats average
also counts the warm-up, parity and file-pipeline launches of a few thousand reads).
    tools/trace_fullsize.py gpurun_out/r03_prof/bench_kernel_trace.csv [regex ...]"""
import csv
import re
import sys


def main():
    path = sys.argv[1]
    pats = [re.compile(p) for p in sys.argv[2:]] or [re.compile(r"gs_match|gs_filter|gi_inflate|gi_segment|gi_find|gi_resolve|gi_crc")]
    dur = {}
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if any(p.search(name) for p in pats):
            dur.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("kernel,calls,full_size_calls,full_size_mean_ms,full_size_min_ms,full_size_max_ms")
    for name, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        big = [x for x in d if x >= 0.8 * max(d)]
        print('"%s",%d,%d,%.4f,%.4f,%.4f' % (name, len(d), len(big), sum(big) / len(big) / 1e6, min(big) / 1e6, max(big) / 1e6))


if __name__ == "__main__":
    main()
