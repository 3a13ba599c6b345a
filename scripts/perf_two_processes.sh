#!/bin/bash
# is the lock-step rate of ONE process limited by the process (runtime, threads) or by the machine (GPU, PCIe)?  The same 32
# streams as one process (4 groups x 4 threads) and as two processes side by side (2 groups x 4 threads each).
run() { python bench.py --workload frame --streams $1 --stream-groups $2 --stream-workers $3 --steps 3000 --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); l=d['lockstep']; print('S=%d groups=%d threads/group=%d: %.0f frames/s' % (l['streams'], l['groups'], l['host_threads_per_group'], l['frames_per_s']))"; }
echo "one process:"; run 32 4 4
echo "two processes side by side:"; run 16 2 4 & run 16 2 4 & wait
echo "four processes side by side:"; run 8 1 4 & run 8 1 4 & run 8 1 4 & run 8 1 4 & wait
