#!/bin/bash
# lock-step front end, 32 streams: steady-state phases of group 0's thread for several (groups : threads per group) shapes
# usage: perf_lockstep_shapes.sh "1:16 2:8 3:5" [streams]
set -e
S=${2:-32}
out=gpurun_out/r05_lockstep_shapes.txt
mkdir -p gpurun_out; : > $out
for shape in $1; do
  g=${shape%%:*}; w=${shape##*:}
  echo "== S=$S groups=$g threads/group=$w $NOTE" >> $out
  python bench.py --workload frame --streams $S --stream-groups $g --stream-workers $w --steps 300 --warmup 100 --no-secondary --no-cpu-baseline 2>>gpurun_out/r05_lockstep_shapes.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['lockstep']
print(json.dumps({k:l[k] for k in ('frames_per_s','ms_per_round','device_waits_ms_per_round_group0','round_phase_ms_mean_group0')}))" >> $out
done
cat $out
