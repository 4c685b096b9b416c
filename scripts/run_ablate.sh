# developer ablation timings of the screened scan (needs a -DPG_SCAN_VARIANTS build); results are wrong by design
# VAR: 0 product, 1 no screen test, 2 no MFMA and no test (stream only)
for b in ${BATCHES:-256}; do
for v in 0 1 2; do
  PG_SCREEN_VAR=$v PG_DEBUG_SCAN=1 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline --latency-reqs 0 2>&1 | grep "plan 0 scan launch 2" | tail -1 | sed "s/^/batch $b var $v: /"
done; done
