"""Developer aid: a few recall passes at one batch size for rocprofv3 --kernel-trace --stats (GPU box):
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 scripts/dev/i4m_prof.py 32"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from oracle import oracle as o

R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
gauss = len(sys.argv) > 3 and sys.argv[3] == "gauss"
loop = len(sys.argv) > 3 and sys.argv[3] == "loop"          # (scripts/dev/power_probe.sh: ~8 s of back-to-back passes)
ctx = pa.Context(0)
t = pa.Table(ctx, rows, 128)
if gauss:
    t.fill_gaussian(o.SEED_TABLE, 1.0)
else:
    t.fill_synthetic(o.SEED_TABLE)
K = 5000
q = o.synth_rows(o.SEED_QUERY, 0, 4096, 128)
d_q = ctx.to_device(q)
d_rows = ctx.malloc(256 * K * 8)
d_sc = ctx.malloc(256 * K * 4)
for i in range(12):
    t.recall_topk_dev(d_q + (i % 8) * 256 * 128 * 4, 256, K, d_rows, d_sc)
if os.environ.get("PG_SWEEP_DEBUG"):
    ctx.set_option("debug_scan", "1")
import time
t_loop = time.perf_counter()
for it in range(1_000_000 if loop else 10):
    if loop and time.perf_counter() - t_loop > 10.0:
        break
    t.recall_topk_dev(d_q + ((it * 7 + 3) % 15) * 256 * 128 * 4, R, K, d_rows, d_sc)
    if not loop or it % 500 == 0:
        ctx.synchronize()
        print(ctx.last_scan_kernel())
