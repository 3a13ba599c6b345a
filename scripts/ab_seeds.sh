#!/bin/bash
# A/B of the seed update on one box: scripts/ab_seeds.sh <lib> [<lib> ...]; "product" = the in-tree library
for L in "$@"; do
  if [ "$L" = product ]; then unset SVOH_LIB; else export SVOH_LIB=$L; fi
  for B in ${SEED_BATCHES:-64}; do
    python bench.py --workload seeds --problems $B --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L B=$B kernel_ms %.4f ms_per_step %.4f' % (d['kernel_ms'], d['ms_per_step']))"
  done
done
