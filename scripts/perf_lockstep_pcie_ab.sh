#!/bin/bash
# What the images' way over PCIe costs the lock-step front end: bench.py --streams 32 as it is (every stream its own copy of the
# sequence in page-locked memory) against SVOH_LOCKSTEP_SHARED_IMAGES=1 (all streams read one copy: the device's caches serve it).
# Same box, alternating, twice.  Output: gpurun_out/r05_lockstep_pcie_ab.txt
set -e
out=gpurun_out/r05_lockstep_pcie_ab.txt
mkdir -p gpurun_out; : > $out
for rep in 1 2; do
  for shared in 0 1; do
    echo "== rep $rep SVOH_LOCKSTEP_SHARED_IMAGES=$shared" >> $out
    SVOH_LOCKSTEP_SHARED_IMAGES=$shared python bench.py --workload frame --streams 32 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>>gpurun_out/r05_lockstep_pcie_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['lockstep']
print(json.dumps({k:l[k] for k in ('frames_per_s','ms_per_round','groups','host_threads_per_group','device_waits_ms_per_round_group0','round_phase_ms_mean_group0')}))" >> $out
  done
done
cat $out
