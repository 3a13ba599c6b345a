#!/bin/bash
# seed-update kernel time with one lane per unit vs eight lanes per unit over the batch size
for B in 2 4 8 16 32 64; do
  for G in 0 1; do
    SVOH_MATCHER_G8=$G timeout -k 10 200 python bench.py --workload seeds --problems $B --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B seeds=%d G8=$G kernel %.3f ms  %.3g seed updates/s' % ($B*3000, d['kernel_ms'], d['value']))" || exit 1
  done
done
