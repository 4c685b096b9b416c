cd /tmp && export TMPDIR=/tmp OMP_WAIT_POLICY=PASSIVE
R=$GRAFT_REPO_ROOT
for n in 32 8; do
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$n -o tl -- python3 $R/scripts/dev/step_timeline.py $n > $R/gpurun_out/tl_$n.log 2>&1
python3 $R/scripts/dev/step_timeline.py --parse $R/gpurun_out/tl_$n > $R/gpurun_out/tl_$n.txt 2>&1
tail -3 $R/gpurun_out/tl_$n.log
done
