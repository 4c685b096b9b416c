"""Developer probe: squared-Euclidean recall (pg_recall_topk_l2) at the benchmark's table shape."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 128, 5000
t = pa.Table(ctx, n, d)
for kind in ("gaussian N(0,1) rows", "normalised rows"):
  if kind.startswith("gauss"):
    t.fill_gaussian(3, 1.0)
  else:
    t.fill_synthetic(o.SEED_TABLE)
  print(kind, "shadow elem bytes", t.screen_info()[0], flush=True)
  for nq in (1, 32, 64, 128, 256):
    q = np.random.default_rng(nq).standard_normal((nq, d)).astype(np.float32)
    if not kind.startswith("gauss"):
        q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    t.recall_topk_l2(q, k)
    best = 1e9
    for _ in range(3):
        t0 = time.time(); rows, dist, cnt = t.recall_topk_l2(q, k); best = min(best, time.time() - t0)
    ms, b = ctx.last_scan_kernel()
    print(f"  l2 nq={nq}: {best*1e3:.2f} ms per call (scan launches {ms:.2f} ms, {b/1e9:.1f} GB streamed)", flush=True)
# sanity on a slice against the oracle
m = 200_000
tab = t.download(0, m)
ts = pa.Table(ctx, m, d); ts.upload(tab)
q = np.random.default_rng(9).standard_normal((5, d)).astype(np.float32)
rows, dist, _ = ts.recall_topk_l2(q, 100)
orow, od = o.recall_topk_l2(tab, q, 100)
print("slice matches oracle:", bool(np.array_equal(rows, orow) and np.array_equal(dist.view(np.uint32), od.view(np.uint32))))
