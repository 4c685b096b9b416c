#!/bin/bash
# developer aid: the batched DPP stage per kernel, matrix pipe (default) against vector pipe (PG_DPP_VALU=1); kernel-trace stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dppm -o t -- python3 $R/scripts/dev/dpp_batch.py > $R/gpurun_out/dppm.log 2>&1
PG_DPP_VALU=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dppv -o t -- python3 $R/scripts/dev/dpp_batch.py > $R/gpurun_out/dppv.log 2>&1
tail -n 2 $R/gpurun_out/dppm.log; tail -n 2 $R/gpurun_out/dppv.log
