"""Developer probe: a big table in ASCENDING score order (every later row beats every earlier one, for all queries) — what the
plans cost when the sample-based thresholds are useless; and the same rows shuffled."""
import os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000, 128, 5000
chunk = 2_000_000
rng = np.random.default_rng(3)
v = rng.standard_normal(d).astype(np.float32); v /= np.linalg.norm(v)
t = pa.Table(ctx, n, d)
for mode in (os.environ.get("MODES", "ascending,shuffled").split(",")):
    perm = None if mode == "ascending" else rng.permutation(n // chunk)
    for c in range(n // chunk):
        src = c if perm is None else int(perm[c])
        scale = np.linspace(0.2 + 0.8 * src * chunk / n, 0.2 + 0.8 * (src + 1) * chunk / n, chunk, dtype=np.float32)[:, None]
        rows = v[None] * scale + 0.002 * rng.standard_normal((chunk, d)).astype(np.float32)
        if perm is not None:
            rows = rows[rng.permutation(chunk)]
        t.upload(np.ascontiguousarray(rows, dtype=np.float32), c * chunk)
    for nq in (1, 128, 256):
        q = (v[None] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
        r0 = ctx.stats().recall_rescans
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); rows_, sc, cnt = t.recall_topk(q, k); ts.append((time.perf_counter() - t0) * 1e3)
        ctx.set_option("recall_exact", "1")
        er, es, _ = t.recall_topk(q[:2], k)
        te = []
        for _ in range(2):
            t0 = time.perf_counter(); t.recall_topk(q, k); te.append((time.perf_counter() - t0) * 1e3)
        ctx.set_option("recall_exact", "0")
        ok = np.array_equal(rows_[:2], er) and np.array_equal(sc[:2].view(np.uint32), es.view(np.uint32))
        print(f"{mode:10s} {n} rows, {nq:3d} queries, K = {k}: calls (ms) " + " ".join("%.1f" % x for x in ts) +
              f"; re-plans {ctx.stats().recall_rescans - r0}; equals the exact scan {ok}; exact scan alone {min(te):.1f} ms", flush=True)
