import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k, nq = 300017, 64, 1000, 7
t = pa.Table(ctx, n, d); t.fill_synthetic(o.SEED_TABLE)
ref = o.synth_rows(o.SEED_TABLE, 0, n, d)
q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
orow, osc = o.recall_topk(ref, q, k)
bad = 0
for it in range(40):
    rows, scores, cnt = t.recall_topk(q, k)
    if not np.array_equal(rows, orow):
        bad += 1
print("mismatching runs:", bad, "of 40")
