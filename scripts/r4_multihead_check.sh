#!/bin/bash
set -x
python -m pytest tests/test_gpu_multihead.py -x -q -m gpu 2>&1 | tail -25
python -m pytest tests/test_gpu_parity.py tests/test_gpu_coalescer.py tests/test_gpu_scene_coalescer.py -x -q -m gpu -k "rank or dnn3 or coalescer or scene" 2>&1 | tail -8
python bench.py --rows 20000000 --steps 6 --warmup 2 --no-extras --no-cpu-baseline --latency-reqs 0 > gpurun_out/r4_mh_bench.json 2> gpurun_out/r4_mh_bench.err; echo rc=$?
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4_mh_bench.json') if l.startswith('{')][-1])
print(json.dumps(d.get('multi_output_rank'), indent=1)); print(d['value'], d['ms_per_step'])
print([ (e['shape'], round(e['ms_per_1280000_items'],4)) for e in d['rank_shapes']])
PY
tail -5 gpurun_out/r4_mh_bench.err
