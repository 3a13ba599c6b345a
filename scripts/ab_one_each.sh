#!/bin/bash
# one 180-patch alignment problem: the problem index from the block index (product) against the work queue (build/libsvo_hip_noeach.so)
for rep in 1 2 3; do
  for L in product build/libsvo_hip_noeach.so; do
    if [ "$L" = product ]; then unset SVOH_LIB; else export SVOH_LIB=$PWD/$L; fi
    python scripts/perf_iter_slope.py 2>&1 | grep -- "->"
  done
done
