#!/bin/bash
# A/B of the alignment kernel's per-pass set-up (LDS level descriptors, vote folded into the reduction): batch and single problem
for L in "$@"; do
  if [ "$L" = product ]; then unset SVOH_LIB; else export SVOH_LIB=$PWD/$L; fi
  echo "== $L"
  ILLUM=0 python scripts/perf_quick.py 2>&1 | grep kernel
  python scripts/perf_iter_slope.py 2>&1 | grep -- "->\|max_iter 16"
done
