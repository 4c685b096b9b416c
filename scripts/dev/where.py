"""Developer probe: a vector recall with a WhereClause (pg_recall_topk_where) at the benchmark's table shape, by selectivity."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 128, 5000
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
feats = pa.Features(ctx, n)
col = np.random.default_rng(1).integers(0, 1_000_000, n).astype(np.int32)
feats.set_column("create_time", pa.F_I32, col)
FRACS = [float(x) for x in os.environ.get('FRACS', '1.0,0.5,0.2,0.1,0.08,0.04,0.01,0.001,0.00003').split(',')]
for frac in FRACS:
    value = int(1_000_000 * (1 - frac))
    for nq in (1, 16, 128):
        q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
        r0 = ctx.stats().recall_rescans
        t.recall_topk_where(feats, "create_time", ">=", value, q, k)
        ts = []
        for _ in range(7):
            t0 = time.time(); rows, sc, cnt = t.recall_topk_where(feats, "create_time", ">=", value, q, k); ts.append(time.time() - t0)
        ts.sort()
        print(f"admitted {frac:8.5f} nq={nq:3d}: min {ts[0]*1e3:7.2f} median {ts[3]*1e3:7.2f} max {ts[-1]*1e3:7.2f} ms per call, count {int(cnt[0])}, re-plans {ctx.stats().recall_rescans - r0}", flush=True)
for l2 in ((True,) if not os.environ.get('NO_L2') else ()):
    for frac in (1.0, 0.5, 0.2, 0.05, 0.001):
        value = int(1_000_000 * (1 - frac))
        for nq in (1, 128):
            q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
            t.recall_topk_where(feats, "create_time", ">=", value, q, k, l2=True)
            ts = []
            for _ in range(7):
                t0 = time.time(); rows, sc, cnt = t.recall_topk_where(feats, "create_time", ">=", value, q, k, l2=True); ts.append(time.time() - t0)
            ts.sort()
            print(f"squared Euclidean, admitted {frac:8.5f} nq={nq:3d}: min {ts[0]*1e3:7.2f} median {ts[3]*1e3:7.2f} max {ts[-1]*1e3:7.2f} ms per call, count {int(cnt[0])}", flush=True)
