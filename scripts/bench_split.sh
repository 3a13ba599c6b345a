#!/bin/bash
# Patch-split Gauss-Newton on the one GPU of a gpurun box: world 1 with a one-rank RCCL group (launch floors),
# then a 2-rank rehearsal on the same device with gloo (correctness of the multi-rank path; its all-reduce time is
# gloo-over-host, not xGMI).  Usage: scripts/bench_split.sh  -> gpurun_out/split_*.json
set -e
mkdir -p gpurun_out
for N in 2000 20000 200000; do
  timeout -k 10 240 python bench.py --workload align-split --features $N --steps 5 --warmup 2 > gpurun_out/split_w1_N$N.json 2> gpurun_out/split_w1_N$N.err
done
SVOH_BENCH_BACKEND=gloo SVOH_BENCH_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --workload align-split --features 20000 --steps 3 --warmup 1 \
  > gpurun_out/split_w2_gloo_N20000.json 2> gpurun_out/split_w2_gloo_N20000.err
