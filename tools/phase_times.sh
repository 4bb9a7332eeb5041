#!/bin/bash
# wave cycles per phase of the short-read match path (developer tool; run on the GPU box from the repo root):
#   tools/phase_times.sh [bench|miss|large|huge]
# builds a copy of libgsgpu.so with -DGS_PHASE=1 (s_memtime stamps at the phase boundaries, accumulated per wave in LDS),
# runs tools/kernel_one.py against it and prints the share of every phase
set -e
cd "$(dirname "$0")/.."
what=${1:-bench}
rm -rf /tmp/gs_phase && mkdir -p /tmp/gs_phase
cp -r genestrip_amd include /tmp/gs_phase/
cp -r tools /tmp/gs_phase/
make -s -C /tmp/gs_phase/genestrip_amd/csrc clean >/dev/null 2>&1 || true
make -s -C /tmp/gs_phase/genestrip_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -DGS_PHASE=1" ../libgsgpu.so ../libgssynth.so
cd /tmp/gs_phase
python3 - "$what" <<'PY'
import ctypes, runpy, sys
sys.argv = ["kernel_one.py", sys.argv[1]]
sys.path.insert(0, "/tmp/gs_phase")
import genestrip_amd as ga
L = ga.lib()
out = (ctypes.c_ulonglong * 16)()
runpy.run_path("/tmp/gs_phase/tools/kernel_one.py", run_name="__main__")
L.gs_debug_phase(out, 0)
names = ["0 offsets", "1 bases -> planes", "2 funnel / act", "3 minimizers (LDS)", "4 orient + gate word", "5 (end of the record path)", "6 contigs + nodes", "7 classify + stats + loop"]
names += ["8 records, sub-round 0", "9 table walk, sub-round 0", "10 records, sub-round 1", "11 table walk, sub-round 1"]
tot = sum(out[i] for i in range(12))
for i, n in enumerate(names):
    print("%-28s %6.2f %%   %8.0f cycles / read" % (n, 100.0 * out[i] / tot, out[i] / 3.0 / 10e6))
print("total wave cycles per read: %.0f (3 launches of 10 M reads)" % (tot / 3.0 / 10e6))
PY
