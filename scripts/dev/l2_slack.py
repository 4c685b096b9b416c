"""Developer probe: screened vs exact squared-Euclidean pass as the rows' norms spread (which l2_max_slack pays?)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k = 20_000_000, 128, 5000
t = pa.Table(ctx, n, d)
rng = np.random.default_rng(1)
for jit in (0.0, 0.02, 0.05, 0.1, 0.2):
    t.fill_synthetic(o.SEED_TABLE)
    if jit:
        for r0 in range(0, n, 1_000_000):
            rows = t.download(r0, 1_000_000)
            rows *= rng.uniform(1 - jit, 1 + jit, (1_000_000, 1)).astype(np.float32)
            t.upload(rows, r0)
    for nq in (1, 32, 128):
        q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
        res = []
        for mode in ("screened", "exact"):
            ctx.set_option("l2_max_slack", "100" if mode == "screened" else "-1")
            t.recall_topk_l2(q, k)
            best = 1e9
            for _ in range(3):
                t0 = time.time(); t.recall_topk_l2(q, k); best = min(best, time.time() - t0)
            res.append(best * 1e3)
        ctx.set_option("debug_scan", "1"); ctx.set_option("l2_max_slack", "100")
        print(f"jitter +-{jit*100:.0f}% nq={nq}: screened {res[0]:.2f} ms, exact {res[1]:.2f} ms", flush=True)
        ctx.set_option("debug_scan", "0")
