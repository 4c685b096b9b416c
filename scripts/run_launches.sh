# per-launch scan timings (plan, launch index, ms) of one bench step at several batch sizes
for b in "$@"; do
  PG_DEBUG_SCAN=1 python bench.py --steps 2 --warmup 1 --batch $b --no-cpu-baseline --latency-reqs 0 2>&1 | grep "pg\]" | tail -${TAILN:-6} | awk -v b=$b '{printf "b%s L%s %s | ", b, $6, $7} END {print ""}'
done
