# developer ablation timings of the 256-query screen (needs a build with SCAN_EXTRA=-DPG_SCAN_VARIANTS); results are wrong by design
# (round 5: such a build is libpairec_gpu_dev.so — export PG_LIB_VARIANT=dev for the runs below)
# VAR: 0 product, 4 test but never the hit path, 1 no screen test, 2 no MFMA and no test (stream only); full pass = launch 2 of the
# pilot plan without the refinement step (includes decode + re-scoring of whatever was staged)
for v in ${VARS:-0 4 1 2}; do
  for e in ${SHARES:-512 604}; do
  echo -n "var $v share $e: "
  PG_NO_PREDICT=1 PG_NO_REFINE=1 PG_SCREEN_EARLY_SHARE=$e PG_SCREEN_VAR=$v PG_DEBUG_SCAN=1 python bench.py --steps 4 --warmup 2 --calibrate 0 --no-cpu-baseline --latency-reqs 0 --no-extras --no-rank-shapes --callers 0 --contexts 1 2>&1 | grep "plan 0 scan launch 2" | tail -3 | awk '{printf "%s ", $(NF-1)}'; echo
  done
done
