"""Developer probe: per-call times, in order, of repeated recalls of one shape (looking for periodic slow calls)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k = 100_000_000, 128, 5000
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
feats = pa.Features(ctx, n)
col = np.random.default_rng(1).integers(0, 1_000_000, n).astype(np.int32)
feats.set_column("create_time", pa.F_I32, col)
def series(name, f, reps=24):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, " ".join("%.1f" % x for x in ts), flush=True)
for nq in (1, 16, 128):
    q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    series(f"plain nq={nq}:", lambda: t.recall_topk(q, k))
    series(f"where 50% nq={nq}:", lambda: t.recall_topk_where(feats, "create_time", ">=", 500_000, q, k))
    series(f"where 1% nq={nq}:", lambda: t.recall_topk_where(feats, "create_time", ">=", 990_000, q, k))
    series(f"l2 plain nq={nq}:", lambda: t.recall_topk_l2(q, k))
