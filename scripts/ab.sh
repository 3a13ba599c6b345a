#!/bin/bash
# A/B of the alignment kernel on one box: scripts/ab.sh <lib> [<lib> ...]; "product" = the in-tree library; P = 4 and 8
for L in "$@"; do
  if [ "$L" = product ]; then unset SVOH_LIB; else export SVOH_LIB=$L; fi
  python scripts/perf_quick.py 2>&1 | grep kernel
  P=8 ILLUM=0 python scripts/perf_quick.py 2>&1 | grep kernel | sed 's/^/P8 /'
done
