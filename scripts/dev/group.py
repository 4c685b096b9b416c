"""Shard-group step on ONE device with logical shards (orchestration overhead check): 100 M x 128 rows in S shards,
256 requests, K = 5000, DPP on the top 500, page 100 — against the single-context pipeline's time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pairec_amd as pa           # noqa: E402
from oracle import oracle as o    # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
w = o.Dnn3Weights()
blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
ex = pa.Expr("${gpu_dnn}*(1+${current_score})^0.1")
q = o.synth_rows(o.SEED_QUERY, 0, 256, 128)
for shards in (1, 2, 4):
    g = pa.ShardGroup([0] * shards)
    g.table_create(rows, 128)
    g.table_fill_synthetic(o.SEED_TABLE)
    g.model_load(pa.MODEL_DNN3, pa.PREC_BF16, blob)
    for dpp in (0, 500):
        g.recommend(ex, "gpu_dnn", q, 5000, 100, dpp_candidates=dpp)      # warm-up (shadows, buffers)
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            out = g.recommend(ex, "gpu_dnn", q, 5000, 100, dpp_candidates=dpp)
        dt = (time.perf_counter() - t0) / n
        print("shards %d dpp %3d: %.2f ms per 256-request step (%.1f M ranked items/s)" % (shards, dpp, dt * 1e3, 256 * 5000 / dt / 1e6))
    g.destroy()
