#!/bin/bash
# Developer aid (GPU box): kernel trace of whole one-in-flight pipeline steps at 1 / 8 / 32 requests; the last step's launches with start
# offsets, durations and gaps -> gpurun_out/tl_<R>.txt (scripts/dev/step_timeline.py)
cd /tmp && export TMPDIR=/tmp OMP_WAIT_POLICY=PASSIVE
R=$GRAFT_REPO_ROOT
for n in 1 8 32; do
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$n -o tl -- python3 $R/scripts/dev/step_timeline.py $n > $R/gpurun_out/tl_$n.log 2>&1
python3 $R/scripts/dev/step_timeline.py --parse $R/gpurun_out/tl_$n > $R/gpurun_out/tl_$n.txt 2>&1
grep "wall ms" $R/gpurun_out/tl_$n.log
done
