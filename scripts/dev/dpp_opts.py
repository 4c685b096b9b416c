"""Developer check of pg_dpp_ex against the oracle, case by case."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pairec_amd as pa
from oracle import oracle as o
ctx = pa.Context(0)
rng = np.random.default_rng(18)
n_tab, d, n, h = 3000, 64, 300, 48
centers = rng.standard_normal((10, d)).astype(np.float32)
tab = (centers[rng.integers(0, 10, n_tab)] + 0.25 * rng.standard_normal((n_tab, d))).astype(np.float32)
t = pa.Table(ctx, n_tab, d)
t.upload(tab)
cand = rng.choice(n_tab, n, replace=False).astype(np.uint32)
rel = np.sort(rng.random(n))[::-1].copy()
hook = rng.standard_normal((n, h))
for has_table, hk, norm, pos, mode, topn, window, alpha in [
        (True, None, True, True, 0, 50, 10, 1.0),
        (True, None, True, True, 1, 50, 10, 1.0),
        (True, None, True, True, 2, 50, 10, 2.0),
        (True, hook, True, True, 0, 40, 7, 1.0),
        (False, hook, True, True, 0, 40, 10, 1.0),
        (False, hook, False, False, 2, 30, 5, 0.05),
        (True, None, False, True, 0, 30, 10, 0.1)]:
    rs, ok = o.dpp_relevance(rel, mode)
    F = o.dpp_features(tab[cand] if has_table else None, hk, norm, pos)
    L = o.dpp_kernel_matrix_f(F, rs, alpha)
    want = o.dpp_with_window(L, topn, window)
    got, used = pa.dpp_ex(ctx, t if has_table else None, cand, rel, alpha, topn, window, norm, pos, mode, hk)
    print((has_table, hk is not None, norm, pos, mode, topn, window), "OK" if np.array_equal(got, want) else "DIFF",
          "rel ok" if np.array_equal(used.view(np.uint64), rs.view(np.uint64)) else "REL DIFF")
    if not np.array_equal(got, want):
        print(" got ", got[:16].tolist(), "\n want", want[:16].tolist())
        print(" diag", np.diag(L)[:12])
