#!/usr/bin/env python3
"""Measured ceilings of the device (gs_calibrate): VALU / SALU issue, vector-load issue, random-line rate.
    python tools/calibrate.py [table MiB ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import genestrip_amd as ga

CLOCK_GHZ = 2.4


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [8, 72, 256, 1152, 9216]
    res = {}
    for name, what in (("valu_pure", ga.CAL_VALU_PURE), ("valu_mix", ga.CAL_VALU_MIX), ("salu", ga.CAL_SALU),
                       ("valu_salu", ga.CAL_VALU_SALU), ("vmem_bytes", ga.CAL_VMEM_BYTES), ("vmem_words", ga.CAL_VMEM_WORDS),
                       ("vmem_shared_lines", ga.CAL_VMEM_SHARED_LINES), ("vmem_scattered", ga.CAL_VMEM_SCATTERED)):
        r = ga.calibrate(what)
        n_cu = r["n_cu"]
        per = r["rate"] / (n_cu * (4 if name.startswith("valu") and name != "valu_salu" else 1))
        r["per_simd_or_cu_per_s"] = per
        r["cycles_at_2.4GHz"] = CLOCK_GHZ * 1e9 / per
        res[name] = r
        print(name, json.dumps(r), flush=True)
    for mib in sizes:
        r = ga.calibrate(ga.CAL_RANDOM_LINES, mib << 20)
        print("random_lines %5d MiB: %.1f G lines/s (%.2f TB/s of 64-byte lines)" % (mib, r["rate"] / 1e9, r["rate"] * 64 / 1e12), flush=True)


if __name__ == "__main__":
    main()
