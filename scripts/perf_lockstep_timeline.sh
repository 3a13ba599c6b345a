#!/bin/bash
# device timelines of the lock-step front end: one group of 11 streams alone, three such groups side by side
set -e
mkdir -p gpurun_out/profiles
for cfg in 11:5:1:5 32:5:3:5 32:15:1:5; do
  MODE=trace LOCKSTEP=$cfg python scripts/profile_chain.py r05 > gpurun_out/r05_timeline_$cfg.log 2>&1
done
cd gpurun_out/profiles
for n in r05_lockstep_S11_W5_G1 r05_lockstep_S32_W5_G3 r05_lockstep_S32_W15_G1; do
  echo "== $n"; python ../../scripts/device_timeline.py ${n}_chain_kernel_trace.csv ${n}_chain_memory_copy_trace.csv 0.4
done > ../r05_lockstep_device_timeline.txt 2>&1
cat ../r05_lockstep_device_timeline.txt
# the traces themselves are large: keep the summary only
rm -f r05_lockstep_S*_chain_kernel_trace.csv r05_lockstep_S*_chain_memory_copy_trace.csv
