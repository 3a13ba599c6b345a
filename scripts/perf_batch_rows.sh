for R in 0 2 4; do for W in 1; do echo "== batch 1024x2000, 512 threads, lanes/patch $R"; SVOH_ALIGN_ROWS=$R SVOH_ALIGN_THREADS=512 ILLUM=0 python scripts/perf_quick.py 2>&1 | grep kernel; done; done
echo "== default (256 threads, staged)"; ILLUM=0 python scripts/perf_quick.py 2>&1 | grep kernel
