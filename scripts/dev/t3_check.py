import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa
ctx = pa.Context(0)
n = 2_000_000
t = pa.Table(ctx, n, 128); t.fill_synthetic(o.SEED_TABLE)
tab_small = o.synth_rows(o.SEED_TABLE, 0, 50000, 128)
w = o.Dnn3Weights()
blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, blob)
rng = np.random.default_rng(1)
sizes = [5000, 1, 0, 333, 32, 33, 31, 64, 65, 127, 257]
users = o.synth_rows(o.SEED_QUERY, 3, len(sizes), 128)
cands = [rng.integers(0, 50000, s_).astype(np.uint32) for s_ in sizes]
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
ref = np.concatenate([o.dnn3_forward(w, 1, users[r], tab_small[cands[r]]) for r in range(len(sizes))])
for knob in (0, 1):
    ctx.set_option("rank_t3", knob)
    got = m.rank_dnn3(t, users, np.concatenate(cands), off)
    print("t3" if knob else "ws", "max |d| vs bf16 oracle", float(np.max(np.abs(got.astype(np.float64) - ref))), flush=True)
# multi-head
wm = o.Dnn3MultiWeights(4)
mm = pa.RankModel(ctx, pa.MODEL_DNN3_MULTI, pa.PREC_BF16, pa.pack_dnn3_multi(wm.w1, wm.b1, wm.w2, wm.b2, wm.w3m, wm.b3m, wm.d_user))
refm = np.concatenate([o.dnn3_multi_forward(wm, 1, users[r], tab_small[cands[r]]) for r in range(len(sizes))], axis=1)
for knob in (0, 1):
    ctx.set_option("rank_t3", knob)
    gm = mm.rank_dnn3(t, users, np.concatenate(cands), off)
    print("multi", "t3" if knob else "ws", float(np.max(np.abs(gm.astype(np.float64) - refm))), flush=True)
R, K = 256, 5000
nI = R * K
cand = rng.integers(0, n, nI).astype(np.uint32)
offs = (np.arange(R + 1) * K).astype(np.uint32)
us = o.synth_rows(o.SEED_QUERY, 0, R, 128)
d_u, d_c, d_o = ctx.to_device(us), ctx.to_device(cand), ctx.to_device(offs)
d_out = ctx.malloc(nI * 4 * 8)
for model, name in ((m, "1 head"), (mm, "4 heads")):
    for knob in (0, 1, 0, 1):
        ctx.set_option("rank_t3", knob)
        best = 1e9
        for it in range(4):
            ctx.synchronize(); t0 = time.time()
            for _ in range(10):
                model.rank_dnn3_dev(t, d_u, d_c, d_o, R, nI, d_out)
            ctx.synchronize(); best = min(best, (time.time() - t0) / 10)
        print(name, "t3" if knob else "ws", "%.4f ms" % (best * 1e3), flush=True)
