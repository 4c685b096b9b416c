cd /tmp && export TMPDIR=/tmp
for v in abl1 abl2; do
  PG_LIB_PATH=$GRAFT_REPO_ROOT/pairec_amd/libpairec_gpu_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dpp_$v -o t -- python3 $GRAFT_REPO_ROOT/scripts/dev/dpp_batch.py > /dev/null 2>&1
done
