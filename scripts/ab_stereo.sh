#!/bin/bash
# A/B of the stereo seam on one box: scripts/ab_stereo.sh <lib> [<lib> ...]; "product" = the in-tree library
for L in "$@"; do
  if [ "$L" = product ]; then unset SVOH_LIB; else export SVOH_LIB=$L; fi
  python bench.py --workload stereo --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L kernel_ms %.4f ms_per_step %.4f' % (d['kernel_ms'], d['ms_per_step']))"
done
