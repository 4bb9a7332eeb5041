#!/bin/bash
# Build ablation variants of libgsgpu.so into build/ablate/ (developer tool, build container):
#   tools/ablate.sh [bits ...] -> libgsgpu_a0.so (reference), _a1 (no per-read reduce), _a2 (no probe), _a3 (neither), _a7 (no gate), _a8 (no statistics atomics)
# then on the GPU box: python tools/stream_times.py build/ablate/libgsgpu_a*.so
set -e
cd "$(dirname "$0")/../genestrip_amd/csrc"
mkdir -p ../../build/ablate
for a in ${@:-0 1 2 3 7}; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value \
        -munsafe-fp-atomics -DGS_ABLATE=$a -shared -o ../../build/ablate/libgsgpu_a$a.so gs_kernels.hip gs_text.hip gs_merge.hip gs_api.cpp -ldl &
done
wait
ls -la ../../build/ablate
