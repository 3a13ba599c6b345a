#!/bin/bash
# rocprofv3 kernel-trace/stats of bench.py --workload klt|seeds -> gpurun_out/profiles/<round>_<workload>_*.csv
set -e
ROUND=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
dst=gpurun_out/profiles
mkdir -p $dst
for wl in klt seeds detect pose; do
  out=/tmp/prof_${ROUND}_$wl
  rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $out.log 2>&1 || { tail -20 $out.log; exit 1; }
  grep '^{' $out.log > $dst/${ROUND}_${wl}_bench_under_rocprof.json || true
  st=$(find $out -name "*kernel_stats.csv" | head -1)
  (head -1 $st; grep -E "svoh::" $st) > $dst/${ROUND}_${wl}_kernel_stats_svoh.csv
  cat $dst/${ROUND}_${wl}_kernel_stats_svoh.csv
done
