#!/bin/bash
# lock-step front end, 32 streams: the phases of group 0's thread in the timed rounds, for a short and a long warm-up
set -e
out=gpurun_out/r05_lockstep_steady.txt
mkdir -p gpurun_out; : > $out
for warm in 20 120; do
  for rep in 1 2; do
    echo "== warm-up $warm rounds, rep $rep" >> $out
    python bench.py --workload frame --streams 32 --steps 400 --warmup $warm --no-secondary --no-cpu-baseline $EXTRA 2>>gpurun_out/r05_lockstep_steady.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['lockstep']
print(json.dumps({k:l[k] for k in ('frames_per_s','ms_per_round','groups','host_threads_per_group','device_waits_ms_per_round_group0','round_phase_ms_mean_group0')}))" >> $out
  done
done
cat $out
