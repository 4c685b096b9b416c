"""Filtered recall (pg_recall_topk_where) under the profiler: 100 M x 128, K = 5000, 128 and 1 queries per call, four selectivities,
6 calls each (scripts/profile_where.sh)."""
import sys, time, os
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # the oracle's OpenMP workers spin after making the queries: keep them off the cores
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
for frac in (0.5, 0.1, 0.04, 0.01):
    for nq in (1, 128):
        q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); t.recall_topk_where(feats, "create_time", ">=", int(1_000_000 * (1 - frac)), q, k); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"where admitted {frac} nq {nq}: calls (ms) " + " ".join("%.2f" % x for x in ts), flush=True)
