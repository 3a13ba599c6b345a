#!/bin/bash
# rocprofv3 --kernel-trace --stats of the lock-step benches (which kernels a round launches, how often, how long): kernel stats CSVs -> gpurun_out/profiles/
# usage: scripts/profile_lockstep_kernels.sh <round>
set -e
ROUND=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
dst=gpurun_out/profiles
mkdir -p $dst
run() {   # tag, bench arguments
  tag=$1; shift
  out=/tmp/prof_ls_$tag; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py "$@" --no-cpu-baseline --no-secondary > $out/log 2>&1 || { tail -20 $out/log; exit 1; }
  grep '^{' $out/log | tail -1 > $dst/${ROUND}_${tag}_bench_under_rocprof.json
  st=$(find $out/trace -name "*kernel_stats.csv" | head -1)
  cp $st $dst/${ROUND}_${tag}_kernel_stats.csv
  head -16 $st | cut -c1-170
}
run lockstep_S32 --workload frame --streams 32 --steps 200 --warmup 50
run lockstep_S32_mix --workload frame --streams 32 --stream-mix --steps 200 --warmup 50
run lockstep_stereo_S32 --workload frame --stereo --streams 32 --steps 150 --warmup 40
echo done
