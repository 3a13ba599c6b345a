#!/bin/bash
# phase stamps of the packed seed kernel: rebuilds matcher.o with -DSVOH_PK_STAMPS in a scratch copy, runs the seeds bench
set -e
cd $GRAFT_REPO_ROOT
rm -rf /tmp/stamps && mkdir -p /tmp/stamps && cp -r svo_pro_universal_amd include oracle bench.py /tmp/stamps/
cd /tmp/stamps/svo_pro_universal_amd/csrc && rm -f matcher.o && make -s EXTRA=-DSVOH_PK_STAMPS > /dev/null 2>&1
cd /tmp/stamps && SVOH_MATCHER_G8=2 python bench.py --workload seeds --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline']['counters']; n=64*3000
print('kernel_ms', d['kernel_ms']); print('avg cycles per lane: search %.0f  setup %.0f  job loop (all lanes avg) %.0f  total %.0f' % tuple(16.0*x/n for x in c))"
cd /tmp/stamps/svo_pro_universal_amd/csrc && rm -f matcher.o && make -s EXTRA="-DSVOH_PK_STAMPS -DSVOH_SEED_STAMPS" > /dev/null 2>&1
cd /tmp/stamps && SVOH_MATCHER_G8=2 python bench.py --workload seeds --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline']['counters']; n=64*3000
print('avg cycles per lane: geometry %.0f  warp %.0f  scan %.0f  set-up %.0f' % tuple(16.0*x/n for x in c))"
