#!/bin/bash
# Round 4: the stationary-weights item-record kernel (csrc/rank_ir.hip) — parity tests, then the cfg-4 leg alone
set -x
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recalls.py tests/test_gpu_scene_coalescer.py tests/test_gpu_coalescer.py -x -q -m gpu -k "fm2t or item_rec or irows or two_tower or fm_" 2>&1 | tail -8
timeout 600 python scripts/dev/cfg4c.py 2>&1 | tail -12
PG_RANK_NO_WS=1 timeout 600 python scripts/dev/cfg4c.py 2>&1 | tail -4
