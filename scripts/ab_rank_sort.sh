for b in 2 3 4 6 8 12 16; do for m in 0 64; do
  PG_RANK_SORT_MAX=$m python bench.py --batch $b --steps 40 --warmup 5 --no-extras --no-cpu-baseline --latency-reqs 0 --contexts 1 2>/dev/null | B=$b M=$m python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch', os.environ['B'], 'rank_sort_max', os.environ['M'], round(d['ms_per_step'],4), 'ms/step')"
done; done
