"""Developer probe: scan-kernel throughput vs table shape."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
for (n, d) in [(200_000_000, 64), (100_000_000, 128), (50_000_000, 256)]:
    t = pa.Table(ctx, n, d); t.fill_synthetic(o.SEED_TABLE)
    for nq in (1, 32):
        q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
        best = 1e9
        for it in range(4):
            t.recall_topk(q, 5000)
            ms, b = ctx.last_scan_kernel(); best = min(best, ms)
        print(f"n={n} d={d} nq={nq}: scan {best:.3f} ms -> {b/best/1e9:.2f} TB/s", flush=True)
    t.destroy()
