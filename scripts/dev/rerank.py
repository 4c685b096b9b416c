"""Wall-clock of the diversity re-rank calls (host buffers in, indices out) at the cfg-5 shape:
500 candidates x 128 dims, page of 100."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from oracle import oracle as o

ctx = pa.Context(0)
n_tab, d, n = 100000, 128, 500
t = pa.Table(ctx, n_tab, d)
t.fill_synthetic(o.SEED_TABLE)
rng = np.random.default_rng(0)
cand = rng.choice(n_tab, n, replace=False).astype(np.uint32)
rel = np.sort(rng.random(n))[::-1].copy()
for name, fn in (("dpp topn=100 window=10", lambda: pa.dpp(ctx, t, cand, rel, 1.0, 100, 10, True)),
                 ("ssd topn=100 window=5", lambda: pa.ssd(ctx, t, cand, rel, 0.25, 100, 5)),
                 ("ssd topn=100 window=10", lambda: pa.ssd(ctx, t, cand, rel, 0.25, 100, 10))):
    fn()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e6)
    print("%-28s p50 %.0f us  min %.0f us" % (name, np.median(ts), min(ts)))
