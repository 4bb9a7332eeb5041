for v in hb6:6 hb7:7 hb8:8 hb8:7 hb8:6; do lib=${v%%:*}; occ=${v##*:};
  echo "== $lib blocks/CU=$occ";
  GS_LIBGSGPU=$PWD/build/var/libgsgpu_$lib.so GS_MATCH_BLOCKS_PER_CU=$occ timeout -k 10 150 python -u tools/kernel_variants.py 2>&1 | grep "^default";
  GS_LIBGSGPU=$PWD/build/var/libgsgpu_$lib.so GS_MATCH_BLOCKS_PER_CU=$occ timeout -k 10 150 python -u tools/miss_heavy.py 2>&1 | grep "miss-only" | cut -c1-60;
  GS_LIBGSGPU=$PWD/build/var/libgsgpu_$lib.so GS_MATCH_BLOCKS_PER_CU=$occ timeout -k 10 300 python -u tools/bench_large.py 25 20 10000000 100000 2>&1 | grep "^match:" ;
done
