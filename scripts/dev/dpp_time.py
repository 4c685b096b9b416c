"""Developer probe: wall time of one pg_dpp call (500 candidates -> 100 picks, window 10) and parity with the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

ctx = pa.Context(0)
n_rows, d = 200_000, 128
t = pa.Table(ctx, n_rows, d)
t.fill_synthetic(o.SEED_TABLE)
tab = o.synth_rows(o.SEED_TABLE, 0, n_rows, d)
rng = np.random.default_rng(3)
for (n, topn, window) in ((500, 100, 10), (500, 50, 16), (1000, 100, 10), (100, 10, 10), (2000, 60, 10)):
    cand = rng.choice(n_rows, n, replace=False).astype(np.uint32)
    rel = np.sort(rng.random(n))[::-1].copy()
    idx = pa.dpp(ctx, t, cand, rel, 1.0, topn, window, True)
    emb = o.l2_normalize_f64(tab[cand].astype(np.float64))
    ref = o.dpp_with_window(o.dpp_kernel_matrix(emb, rel, 1.0), topn, window)
    ts = []
    for _ in range(10):
        t0 = time.time(); pa.dpp(ctx, t, cand, rel, 1.0, topn, window, True); ts.append((time.time() - t0) * 1e3)
    print(f"n={n} topn={topn} window={window}: identical={np.array_equal(idx, ref)} wall min {min(ts):.3f} ms", flush=True)
