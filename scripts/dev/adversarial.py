"""Developer probe: tables ordered so that every later row beats every earlier one for all queries (the safe plan's case), at
256 queries and small K — the hit-record path with every row of a bounded chunk a suspect of every query."""
import sys, os
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
rng = np.random.default_rng(3)
d = 128
for n, k, nq in ((2_000_000, 10, 256), (2_000_000, 1, 200), (1_500_000, 300, 256), (3_000_000, 10, 129)):
    v = rng.standard_normal(d).astype(np.float32); v /= np.linalg.norm(v)
    scale = np.linspace(0.2, 1.0, n, dtype=np.float32)[:, None]
    tab = (v[None, :] * scale + 0.002 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    q = (v[None, :] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    r0 = ctx.stats().recall_rescans
    rows, sc, cnt = t.recall_topk(q, k)
    sel = [0, nq // 2, nq - 1]
    orow, osc = o.recall_topk(tab, q[sel], k)
    ok = np.array_equal(rows[sel], orow) and np.array_equal(sc[sel].view(np.uint32), osc.view(np.uint32))
    print(f"ascending table n={n} k={k} nq={nq}: matches oracle {ok}, re-plans {ctx.stats().recall_rescans - r0}", flush=True)
    t.destroy()
