#!/bin/bash
# sweep of the text-path knobs of gs_host_match_files (developer tool; run on the GPU box)
cd "$(dirname "$0")/.." || exit 1
for r in 4 8 16; do for b in 8 32 128; do
  echo "readers=$r block=${b}MiB"
  GS_HOST_READERS=$r GS_HOST_BLOCK_BYTES=$((b<<20)) timeout -k 10 300 python -u tools/file_rate.py ${1:-16000000} 2>&1 | grep "page cache"
done; done
