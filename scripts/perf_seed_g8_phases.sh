#!/bin/bash
# mean cycles per unit and phase of the eight-lane seed kernel at a per-frame batch size (diagnostic build in a scratch copy)
set -e
cd $GRAFT_REPO_ROOT
rm -rf /tmp/stamps && mkdir -p /tmp/stamps && cp -r svo_pro_universal_amd include oracle bench.py /tmp/stamps/
cd /tmp/stamps/svo_pro_universal_amd/csrc && rm -f matcher.o && make -s EXTRA=-DSVOH_SEED_STAMPS > /dev/null 2>&1
cd /tmp/stamps && SVOH_MATCHER_G8=1 python bench.py --workload seeds --problems ${B:-1} --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline']['counters']; n=int(os.environ.get('B','1'))*3000
print('kernel_ms', d['kernel_ms']); print('mean cycles per unit: geometry+rest %.0f  warp %.0f  scan %.0f  align %.0f' % tuple(16.0*x/n for x in c))"
