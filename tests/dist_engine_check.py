"""Child process of tests/test_gpu_group.py::test_dist_engine_single_rank_matches_oracle (torch initialises first)."""
import os
import sys

import numpy as np
import torch

torch.cuda.set_device(0)
torch.zeros(1, device="cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pairec_amd as pa                                # noqa: E402
from oracle import oracle as o                          # noqa: E402
from pairec_amd.dist import GpuShardEngine, shard_context, sharded_step   # noqa: E402
from test_gpu_group import EXPR, oracle_pipeline       # noqa: E402

n, d, k, R, top_n, dpp_c = 80_000, 128, 300, 6, 25, 90
ctx, _stream = shard_context(torch, pa, 0)
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
w = o.Dnn3Weights()
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
ex = pa.Expr(EXPR)
eng = GpuShardEngine(torch, ctx, t, m, ex, k, R)
q = o.synth_rows(o.SEED_QUERY, 77, R, d)
tq = torch.from_numpy(q).to("cuda:0")
torch.cuda.synchronize()                                # the upload ran on the default stream
rows, fused, order, page = sharded_step(eng, None, torch, tq, R, k, top_n, {"candidates": dpp_c, "alpha": 1.0, "window": 10})
torch.cuda.synchronize()
rows, page = rows.cpu().numpy().astype(np.uint64), page.cpu().numpy().astype(np.int64)
want = oracle_pipeline(tab, w, pa.PREC_F32, q, k, top_n, dpp_c, 1.0, 10)
for r in range(R):
    assert np.array_equal(rows[r][page[r]], want[r][0]), "request %d: page differs" % r
print("dist engine OK")
