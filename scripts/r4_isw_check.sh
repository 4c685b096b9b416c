#!/bin/bash
# Round 4: fm2t_isw_kernel (csrc/rank_is.hip, the default for cfg 4's shape) — the FM + two-tower parity tests on it, then the
# cfg-4 leg on it and on fm2t_irs_kernel (PG_FM2T_IRS=1)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recalls.py tests/test_gpu_scene_coalescer.py tests/test_gpu_coalescer.py -x -q -m gpu -k "fm2t or item_rec or irows or two_tower or fm_" 2>&1 | tail -8
timeout 600 python scripts/dev/cfg4c.py 2>&1 | tail -12 | head -4
timeout 300 python scripts/dev/cfg4_prof.py random 2>&1 | tail -1
PG_FM2T_IRS=1 timeout 300 python scripts/dev/cfg4_prof.py random 2>&1 | tail -1
