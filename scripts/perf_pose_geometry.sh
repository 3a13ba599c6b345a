#!/bin/bash
# Kernel time of the pose optimiser in its two geometries (one wave / four waves per bundle) over the batch size.
# Writes gpurun_out/pose_geometry.txt
out=gpurun_out/pose_geometry.txt
: > $out
for nt in 256 64; do
  for b in 64 256 512 2048; do
    SVOH_POSE_THREADS=$nt timeout -k 10 120 python -u bench.py --workload pose --problems $b --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads=$nt bundles=$b kernel_ms=%.4f bundles/s=%.4g' % (d['kernel_ms'], d['value']))" >> $out || exit 1
  done
done
cat $out
