#!/bin/bash
# N > 1 code paths on a one-GPU box: the child-rank GPU tests, then bench.py's four N > 1 modes in the shared-GPU dev mode.
set -x
mkdir -p gpurun_out/r4_multi
python -m pytest tests/test_gpu_group.py -x -q -m gpu -k "dist_engine" 2>&1 | tail -15
python bench.py --gpus 2 --steps 1; echo "plain --gpus 2 on a 1-GPU box: rc=$?"
export PG_BENCH_SHARE_GPU=1
timeout 600 python bench.py --gpus 2 --mode shard --rows 3000000 --steps 4 --warmup 2 > gpurun_out/r4_multi/shard2.json 2> gpurun_out/r4_multi/shard2.err; echo rc=$?
timeout 600 python bench.py --gpus 2 --mode replica --rows 3000000 --steps 4 --warmup 2 > gpurun_out/r4_multi/replica2.json 2> gpurun_out/r4_multi/replica2.err; echo rc=$?
timeout 600 python bench.py --gpus 4 --mode group --rows 3000000 --steps 4 --warmup 2 > gpurun_out/r4_multi/group4.json 2> gpurun_out/r4_multi/group4.err; echo rc=$?
timeout 600 python bench.py --gpus 2 --mode router --rows 3000000 --steps 4 --warmup 2 --callers 256 > gpurun_out/r4_multi/router2.json 2> gpurun_out/r4_multi/router2.err; echo rc=$?
unset PG_BENCH_SHARE_GPU
timeout 600 python bench.py --gpus 1 --mode group --rows 20000000 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4_multi/group1.json 2> gpurun_out/r4_multi/group1.err; echo rc=$?
timeout 600 python bench.py --gpus 1 --mode router --rows 20000000 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4_multi/router1.json 2> gpurun_out/r4_multi/router1.err; echo rc=$?
tail -c 600 gpurun_out/r4_multi/*.err
for f in gpurun_out/r4_multi/*.json; do echo $f; head -c 700 $f; echo; done
