#!/bin/bash
# developer aid: kernel trace of the cfg-4 rank stage alone (scripts/dev/cfg4_prof.py): which launches make up the stage
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg4_trace -o t -- python3 $R/scripts/dev/cfg4_prof.py random sync > $R/gpurun_out/cfg4_trace.log 2>&1
grep -a "device ms" $R/gpurun_out/cfg4_trace.log
