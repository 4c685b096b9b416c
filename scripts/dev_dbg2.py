import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
for (n, d, k, nq) in [(300017,64,5000,7),(300017,64,5000,1),(300017,64,1000,7),(100000,64,5000,7),(40000,64,5000,7),(30000,64,5000,7)]:
    t = pa.Table(ctx, n, d); t.fill_synthetic(o.SEED_TABLE)
    ref = o.synth_rows(o.SEED_TABLE, 0, n, d)
    q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    rows, scores, cnt = t.recall_topk(q, k)
    orow, osc = o.recall_topk(ref, q, k)
    ok = np.array_equal(rows, orow)
    msg = ""
    if not ok:
        for qi in range(nq):
            missing = np.setdiff1d(orow[qi], rows[qi]); extra = np.setdiff1d(rows[qi], orow[qi])
            if len(missing): msg += f"\n   q{qi}: missing={len(missing)} {missing[:6]} blk={np.unique(missing//32)[:6]} extra={extra[:6]} dup_in_result={len(rows[qi])-len(np.unique(rows[qi]))}"
    print(f"n={n} d={d} k={k} nq={nq}: rows_exact={ok}{msg}", flush=True)
    t.destroy()
