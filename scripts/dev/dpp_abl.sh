#!/bin/bash
# developer aid: kernel times of the DPP stage with an ablation build (make -C pairec_amd/csrc DPP_EXTRA=-DDPP_ABL=n → libpairec_gpu_dev.so):
# 1 no stores of S, 2 no matrix instructions, 4 the greedy kernel reads a row that does not depend on its pick (wrong results by design)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PG_LIB_VARIANT=dev rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dpp_abl -o t -- python3 $R/scripts/dev/dpp_batch.py > $R/gpurun_out/dpp_abl.log 2>&1
grep -a "dpp_" $R/gpurun_out/dpp_abl/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
