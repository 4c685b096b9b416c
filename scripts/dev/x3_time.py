"""Developer probe: the rank stage alone (256 x 5 000 random candidates of a big table) per precision mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

n = int(os.environ.get("X3_ROWS", 20_000_000))
R, K = 256, 5000
ctx = pa.Context(0)
t = pa.Table(ctx, n, 128); t.fill_synthetic(o.SEED_TABLE)
w = o.Dnn3Weights()
blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
rng = np.random.default_rng(1)
nI = R * K
cand = rng.integers(0, n, nI).astype(np.uint32)
offs = (np.arange(R + 1) * K).astype(np.uint32)
us = o.synth_rows(o.SEED_QUERY, 0, R, 128)
d_u, d_c, d_o = ctx.to_device(us), ctx.to_device(cand), ctx.to_device(offs)
d_out = ctx.malloc(nI * 4)
outs = {}
for prec in [int(x) for x in os.environ.get("X3_PRECS", "1,2,0").split(",")]:
    m = pa.RankModel(ctx, pa.MODEL_DNN3, prec, blob)
    best = 1e9
    for it in range(4):
        ctx.synchronize(); t0 = time.time()
        for _ in range(10):
            m.rank_dnn3_dev(t, d_u, d_c, d_o, R, nI, d_out)
        ctx.synchronize(); dt = (time.time() - t0) / 10
        best = min(best, dt)
    got = np.empty(nI, np.float32); ctx.d2h(got, d_out); outs[prec] = got
    print(f"DNN3 prec={prec}: {best*1e3:.3f} ms/call -> {nI/best/1e6:.1f} M items/s; executed {nI*393216*(3 if prec==2 else 1)/best/1e12:.0f} TFLOP/s", flush=True)
    m.destroy()
if 0 in outs:
    for p in outs:
        if p: print(f"prec {p} vs f32 on device: max |d| {np.max(np.abs(outs[p].astype(np.float64)-outs[0])):.3e}")
