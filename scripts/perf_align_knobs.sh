for K in "" "SVOH_ALIGN_THREADS=512" "SVOH_ALIGN_WG_PER_CU=1" "SVOH_ALIGN_THREADS=512 SVOH_ALIGN_LDS=131072" "SVOH_ALIGN_THREADS=1024"; do
  echo "== $K"; env $K ILLUM=0 python scripts/perf_quick.py 2>&1 | grep kernel
done
