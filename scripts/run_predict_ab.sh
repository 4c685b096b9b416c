#!/bin/bash
# A/B of the threshold model (PG_NO_PREDICT) on the headline measurement, uniform and Gaussian rows (developer aid)
cd "$(dirname "$0")/.."
for dist in uniform gaussian; do
  for np in 0 1; do
    echo -n "table $dist, PG_NO_PREDICT=$np: "
    if [ $np = 1 ]; then export PG_NO_PREDICT=1; else unset PG_NO_PREDICT; fi
    python bench.py --no-extras --no-cpu-baseline --latency-reqs 0 --callers 0 --steps 40 --warmup 8 --table-dist $dist 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f M items/s  %.3f ms/step  scan %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['ms_per_pass']))"
  done
done
unset PG_NO_PREDICT
PG_DEBUG_SCAN=1 python bench.py --no-extras --no-cpu-baseline --latency-reqs 0 --callers 0 --steps 6 --warmup 6 2>&1 >/dev/null | grep "threshold model" | tail -4
