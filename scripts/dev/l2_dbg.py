import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n, d, k = 20_000_000, 128, 5000
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
for nq in (1, 5):
    q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    t.recall_topk_l2(q, k)
    print("---- nq", nq, flush=True)
    ctx.set_option("debug_scan", "1")
    t.recall_topk_l2(q, k)
    ctx.set_option("debug_scan", "0")
