"""Developer probe: recall time vs the rows' distribution (uniform / Gaussian / heavy-tailed), 20 M x 128, 256 queries,
K = 5000 — how much the single table-wide int8 scale costs when the rows' largest elements differ."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa

ctx = pa.Context(0)
n, d, nq, k = 20_000_000, 128, 256, 5000
rng = np.random.default_rng(0)
t = pa.Table(ctx, n, d)
for name in ("uniform", "gauss", "student3"):
    chunk = 1_000_000
    for r0 in range(0, n, chunk):
        if name == "uniform":
            x = rng.uniform(-1, 1, (chunk, d))
        elif name == "gauss":
            x = rng.standard_normal((chunk, d))
        else:
            x = rng.standard_t(3, (chunk, d))
        t.upload(x.astype(np.float32), row0=r0)
    q = (rng.uniform(-1, 1, (nq, d)) if name == "uniform" else rng.standard_normal((nq, d))).astype(np.float32)
    eb, sc, rs = t.screen_info()
    t.recall_topk(q, k)
    t0 = time.perf_counter()
    for _ in range(3):
        t.recall_topk(q, k)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("%-9s shadow %d B/elem scale %.4g resid %.4g  recall %.2f ms (host buffers)" % (name, eb, sc, rs, ms), flush=True)
