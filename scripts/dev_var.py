import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d = 100_000_000, 128
t = pa.Table(ctx, n, d); t.fill_synthetic(o.SEED_TABLE)
q = o.synth_rows(o.SEED_QUERY, 0, 32, d)
for var in ("0", "1", "2", "3", "0"):
    os.environ["PG_SCAN_VAR"] = var
    best = 1e9
    for it in range(4):
        try:
            t.recall_topk(q, 5000)
        except Exception as e:
            pass
        ms, b = ctx.last_scan_kernel(); best = min(best, ms)
    print(f"VAR={var}: scan {best:.3f} ms -> {b/best/1e9:.2f} TB/s", flush=True)
