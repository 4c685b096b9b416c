"""Developer probe: the benchmark's bf16 DNN3 rank stage on the three-waves-per-SIMD kernel (rank_t3) vs rank_ws."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n = 20_000_000
t = pa.Table(ctx, n, 128); t.fill_synthetic(o.SEED_TABLE)
w = o.Dnn3Weights()
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
rng = np.random.default_rng(1)
R, K = 256, 5000
nI = R * K
d_u = ctx.to_device(o.synth_rows(o.SEED_QUERY, 0, R, 128))
d_c = ctx.to_device(rng.integers(0, n, nI).astype(np.uint32))
d_o = ctx.to_device((np.arange(R + 1) * K).astype(np.uint32))
d_out = ctx.malloc(nI * 4)
for knob in [int(x) for x in os.environ.get("T3_KNOBS", "0,1,0,1").split(",")]:
    ctx.set_option("rank_t3", knob)
    best = 1e9
    for it in range(4):
        ctx.synchronize(); t0 = time.time()
        for _ in range(10):
            m.rank_dnn3_dev(t, d_u, d_c, d_o, R, nI, d_out)
        ctx.synchronize(); best = min(best, (time.time() - t0) / 10)
    print("t3" if knob else "ws", "%.4f ms" % (best * 1e3), flush=True)
