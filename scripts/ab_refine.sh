#!/bin/bash
# A/B of the prefix plan / threshold refinement (PG_NO_REFINE=1 = off) at several batch sizes, on one box
for b in ${BATCHES:-8 32 64 128}; do for e in 0 1; do
  if [ $e = 1 ]; then export PG_NO_REFINE=1; else unset PG_NO_REFINE; fi
  python bench.py --batch $b --steps 30 --warmup 5 --no-extras --no-cpu-baseline --latency-reqs 0 2>/dev/null | B=$b python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch', os.environ['B'], 'no_refine', os.environ.get('PG_NO_REFINE'), round(d['value']/1e6,1), 'M items/s', round(d['ms_per_step'],3), 'ms/step', round(d['roofline']['ms_per_pass'],3), 'ms/pass')"
done; done
