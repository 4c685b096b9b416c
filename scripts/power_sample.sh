python bench.py --no-extras --no-cpu-baseline --latency-reqs 0 --callers 0 --no-rank-shapes --steps 3000 --warmup 4 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature" | tr '\n' ';' | cut -c1-600; echo; sleep 1; done
wait $BP
python -c "import json; d=json.load(open('/tmp/b.json')); print(d['value']/1e6, d['ms_per_step'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
