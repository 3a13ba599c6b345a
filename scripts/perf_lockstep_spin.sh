#!/bin/bash
# how the lock-step rate depends on the idle policy of the worker threads and on the number of groups (bench.py --streams S)
S=${1:-32}
for cfg in "20000 400 4:4" "20000 400 8:2" "200 0 4:4" "200 0 8:2" "200 0 8:4" "200 0 6:5" "2000 50 8:4" "0 0 8:4" "200 0 16:2"; do
  set -- $cfg
  echo -n "spin=$1 yield=$2 -> "
  SVOH_LOCKSTEP_SPIN=$1 SVOH_LOCKSTEP_YIELD=$2 scripts/perf_bench_streams.sh $S $3
done
