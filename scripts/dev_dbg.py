import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
for (n, d, k, nq) in [(40000,128,200,1),(70000,128,200,1),(140000,128,200,32),(200000,128,500,7),(200000,64,500,1),(300017,64,5000,7),(1000000,64,200,2),(500000,256,300,3),(123457,192,16384,2)]:
    t = pa.Table(ctx, n, d); t.fill_synthetic(o.SEED_TABLE)
    ref = o.synth_rows(o.SEED_TABLE, 0, n, d)
    q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    rows, scores, cnt = t.recall_topk(q, k)
    orow, osc = o.recall_topk(ref, q, k)
    print(f"n={n} d={d} k={k} nq={nq}: rows_exact={np.array_equal(rows, orow)} scores_bitexact={np.array_equal(scores.view(np.uint32), osc.view(np.uint32))}", flush=True)
    t.destroy()
