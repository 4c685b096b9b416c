#!/bin/bash
# Round 5: full GPU test suite, the headline rocprofv3 summary (kernel trace + PMC), the x3 kernel's PMC summary and the
# default bench line.  Outputs under gpurun_out/ (copied to profiles/ by hand).
cd "$(dirname "$0")/.."
python -m pytest tests -x -q -m gpu > gpurun_out/r5_gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_gputests.log
tail -3 gpurun_out/r5_gputests.log
bash scripts/profile_r4_headline.sh r5_headline > gpurun_out/r5_headline_profile.log 2>&1
X3_PRECS=2 X3_ROWS=20000000 bash scripts/profile_kernel_r5.sh r5_x3 dnn3_x3 scripts/dev/x3_time.py > gpurun_out/r5_x3_profile.log 2>&1
python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo "bench rc $?"
tail -c 600 gpurun_out/r5_bench_default.err
