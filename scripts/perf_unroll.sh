#!/bin/bash
# alignment kernel time for row-unroll settings of patch_moments (diagnostic builds in a scratch copy)
cd $GRAFT_REPO_ROOT
for cfg in "1 4" "1 8" "2 4" "2 8" "1 2"; do
  set -- $cfg
  rm -rf /tmp/ur && mkdir -p /tmp/ur && cp -r svo_pro_universal_amd include oracle bench.py /tmp/ur/
  (cd /tmp/ur/svo_pro_universal_amd/csrc && rm -f sparse_align.o && make -s EXTRA="-DSVOH_ROW_UNROLL=$1 -DSVOH_ROW_UNROLL_GONLY=$2" > /dev/null 2>&1)
  for p in ${PATCHES:-8 4}; do
    (cd /tmp/ur && python bench.py --patch $p --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unroll full=$1 gonly=$2 P', d['config']['patch_size'], 'kernel_ms %.3f' % d['kernel_ms'])")
  done
done
