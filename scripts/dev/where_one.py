"""Developer probe: one filtered recall shape with the scan's debug trace (PG_DEBUG_SCAN=1)."""
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
for spec in sys.argv[1:]:
    frac, nq, l2 = spec.split(":")
    frac, nq, l2 = float(frac), int(nq), int(l2)
    value = int(1_000_000 * (1 - frac))
    q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    t.recall_topk_where(feats, "create_time", ">=", value, q, k, l2=bool(l2))
    ctx.set_option("debug_scan", 1)
    print(f"--- admitted {frac} nq {nq} l2 {l2}", file=sys.stderr, flush=True)
    t0 = time.time(); t.recall_topk_where(feats, "create_time", ">=", value, q, k, l2=bool(l2)); dt = time.time() - t0
    ctx.set_option("debug_scan", 0)
    print(f"admitted {frac} nq {nq} l2 {l2}: {dt*1e3:.2f} ms", file=sys.stderr, flush=True)
