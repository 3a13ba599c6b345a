#!/bin/bash
# cluster mode of the alignment with the shares of a problem on one XCD (product) against queue order (build/libsvo_hip_noxcd.so)
for rep in 1 2; do
  for L in product build/libsvo_hip_noxcd.so; do
    if [ "$L" = product ]; then unset SVOH_LIB; else export SVOH_LIB=$PWD/$L; fi
    echo "## $L"
    NS=2000,8000,20000 REPS=24 python scripts/perf_latency_cluster.py 2>&1 | grep "^N=" | cut -c1-230
  done
done
