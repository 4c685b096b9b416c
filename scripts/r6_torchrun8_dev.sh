#!/bin/bash
# Round 6 (as round 5): the driver's N = 8 launch line on ONE GPU (PG_BENCH_SHARE_GPU=1: every rank on cuda:0, gloo with host-staged payloads):
# the preflight at world size 8 (7 requests: fewer than ranks), the replica headline with its shard / group sub-objects.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6_torchrun8
export PG_BENCH_SHARE_GPU=1
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 8 --steps 3 --warmup 1 --rows 2000000 --calibrate 2 --no-cpu-baseline \
  > gpurun_out/r6_torchrun8/replica_8.json 2> gpurun_out/r6_torchrun8/replica_8.err
echo "rc $?"
tail -c 1500 gpurun_out/r6_torchrun8/replica_8.json
tail -5 gpurun_out/r6_torchrun8/replica_8.err
