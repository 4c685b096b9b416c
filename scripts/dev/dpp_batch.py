"""developer aid: the batched DPP stage alone (pg_dpp_batch_dev, 256 requests x 500 candidates -> 100 picks, window 10)
and pg_ssd alone — what scripts/kstats_py.sh profiles for profiles/r3_dpp_ssd_kernel_stats.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from pairec_amd import _lib
from oracle import oracle as o
ctx = pa.Context(0)
R, n, d, topn, window = 256, 500, 128, 100, 10
rng = np.random.default_rng(1)
emb = rng.standard_normal((R, n, d)).astype(np.float32)
rel = np.sort(rng.random((R, n)), axis=1)[:, ::-1].copy()
d_e, d_r = ctx.to_device(emb), ctx.to_device(rel)
d_o, d_c = ctx.malloc(R * topn * 4), ctx.malloc(R * 4)
def call():
    _lib.check(ctx.L.pg_dpp_batch_dev(ctx.h, d_e, d_r, R, n, d, 1.0, topn, window, 1, d_o, d_c))
for _ in range(3): call()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(10): call()
ctx.synchronize()
print("pg_dpp_batch_dev %d x %d -> %d: %.3f ms per batch" % (R, n, topn, (time.perf_counter() - t0) / 10 * 1e3))
t = pa.Table(ctx, 200_000, d)
t.fill_synthetic(o.SEED_TABLE)
cand = rng.choice(200_000, n, replace=False).astype(np.uint32)
for _ in range(3): pa.ssd(ctx, t, cand, rel[0], 0.25, topn, 5)
t0 = time.perf_counter()
for _ in range(10): pa.ssd(ctx, t, cand, rel[0], 0.25, topn, 5)
print("pg_ssd %d -> %d: %.3f ms per call" % (n, topn, (time.perf_counter() - t0) / 10 * 1e3))
