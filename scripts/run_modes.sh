set -x
python -m pytest tests/test_gpu_parity.py -q -x -k "recall" 2>&1 | tail -3
for m in 1 3; do
PG_SCREEN_MODE=$m python bench.py --steps 10 --warmup 2 --batch 256 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('mode $m', j['value'], j['ms_per_step'], j['roofline'])"
done
