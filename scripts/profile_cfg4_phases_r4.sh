#!/bin/bash
# Round 4: per-phase cycle stamps of fm2t_irs_kernel (csrc/rank_ir.hip built with -DPG_IR_PROFILE; the marks' s_memtime
# reads force lgkmcnt(0), so the profiled kernel is a little slower than the product one).  Run on the GPU box:
#   bash scripts/profile_cfg4_phases_r4.sh > gpurun_out/r4_cfg4_phases.txt 2>&1
set -e
cd "$(dirname "$0")/.."
touch pairec_amd/csrc/rank_ir.hip
make -C pairec_amd/csrc WS_EXTRA=-DPG_IR_PROFILE -j8 > /dev/null
# (the instrumented build is libpairec_gpu_dev.so, selected by PG_LIB_VARIANT=dev: the product library is not replaced)
PG_LIB_VARIANT=dev PG_FM2T_IRS=1 python scripts/dev/cfg4_prof.py random 2>&1 | grep -v "^$" | tail -14
PG_LIB_VARIANT=dev PG_FM2T_IRS=1 python scripts/dev/cfg4_prof.py row0 2>&1 | grep -v "^$" | tail -14
