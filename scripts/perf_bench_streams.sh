#!/bin/bash
# bench.py --workload frame --streams S for several (groups, threads per group): frames/s and ms per round
S=${1:-32}
shift
for gw in "$@"; do
  g=${gw%%:*}; w=${gw##*:}
  python bench.py --workload frame --streams $S --stream-groups $g --stream-workers $w --steps 200 --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); l=d['lockstep']; print('S=%d groups=%d threads/group=%d: %.0f frames/s, %.3f ms per round; group 0 inside its rounds: %s' % (l['streams'], l['groups'], l['host_threads_per_group'], l['frames_per_s'], l['ms_per_round'], {k: round(v,3) for k,v in l['round_stage_ms_median_group0'].items()}))"
done
