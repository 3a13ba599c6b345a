#!/bin/bash
# phase stamps of the alignment kernel (diagnostic build with -DSVOH_PHASE_STAMPS in a scratch copy)
set -e
cd $GRAFT_REPO_ROOT
rm -rf /tmp/stamps && mkdir -p /tmp/stamps && cp -r svo_pro_universal_amd include oracle bench.py scripts /tmp/stamps/
cd /tmp/stamps/svo_pro_universal_amd/csrc && rm -f sparse_align.o && make -s EXTRA="-DSVOH_PHASE_STAMPS $STAMPS_EXTRA" > /dev/null 2>&1
cd /tmp/stamps && B=${B:-1024} P=${P:-4} python scripts/perf_stamps.py 2>&1 | grep -E "stamps|iters" | head -40
