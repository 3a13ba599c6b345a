#!/bin/bash
# frames/s of 32 streams that differ (bench.py --stream-mix) against the number of lock-step groups, threads per group and the
# runtime's hardware queues (GPU_MAX_HW_QUEUES); one line per setting
# (an EMPTY GPU_MAX_HW_QUEUES is not "unset": the runtime then falls back to one queue and everything halves -- measured by accident, round 6)
for q in 4 8 16; do for gw in "4 4" "8 2" "5 3" "6 2" "2 8"; do
  set -- $gw
  GPU_MAX_HW_QUEUES=$q python bench.py --workload frame --streams 32 --stream-mix --stream-groups $1 --stream-workers $2 --steps 300 --warmup 80 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); l=d['lockstep']; print('hwq=%-3s G=$1 W=$2: mix %6.0f frames/s (round %.3f ms, device waits %.3f) | identical %6.0f' % ('$q', d['value'], l['ms_per_round'], l['device_waits_ms_per_round_group0'], (d.get('identical_streams_beside_it') or {}).get('frames_per_s', 0)))"
done; done
