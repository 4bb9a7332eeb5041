#!/bin/bash
# kernel time of the headline workload by resident workgroups per CU (= waves per SIMD) of gs_match_kernel
# (developer tool; run on the GPU box from the repo root): how much of the time is latency that more waves would hide
mkdir -p gpurun_out
for b in 8 7 6 5 4; do
  GS_MATCH_BLOCKS_PER_CU=$b python bench.py --pmc off --legs main,large --cpu-seconds 0 --check-reads 20000 2>/dev/null > gpurun_out/occ_$b.json
  python3 -c "
import json
d=json.loads(open('gpurun_out/occ_$b.json').read().strip().splitlines()[-1]); print('blocks/CU $b', 'configs[1]', d['roofline']['kernel_ms'], 'ms; 47 M store', d['large_store']['kernel_ms'], 'ms')"
done
