#!/bin/bash
# alignment kernel time under a few instruction-scheduler settings (diagnostic builds in a scratch copy)
cd $GRAFT_REPO_ROOT
i=0
while IFS= read -r flags; do
  i=$((i+1))
  rm -rf /tmp/sf && mkdir -p /tmp/sf && cp -r svo_pro_universal_amd include oracle bench.py /tmp/sf/
  (cd /tmp/sf/svo_pro_universal_amd/csrc && rm -f sparse_align.o && make -s EXTRA="$flags" > /tmp/sf/build.log 2>&1) || { echo "[$flags] build failed: $(grep -m1 error /tmp/sf/build.log | cut -c1-160)"; continue; }
  for p in ${PATCHES:-4 8}; do
    (cd /tmp/sf && timeout -k 5 120 python bench.py --patch $p --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$flags] P', d['config']['patch_size'], 'kernel_ms %.3f' % d['kernel_ms'])")
  done
done <<'FLAGS'
-DSVOH_BASELINE_FLAGS
-mllvm -amdgpu-sched-strategy=max-ilp
-mllvm -amdgpu-sched-strategy=max-memory-clause
-mllvm -amdgpu-schedule-metric-bias=0
FLAGS
