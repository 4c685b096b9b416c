"""Developer probe: rank model parity (DNN3 + FM two-tower) and throughput on one MI355X."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

ctx = pa.Context(0)
n, d = 20000, 128
t = pa.Table(ctx, n, d); t.fill_synthetic(o.SEED_TABLE)
tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
print("fill bitexact", np.array_equal(tab.view(np.uint32), t.download(0, n).view(np.uint32)))
w = o.Dnn3Weights()
blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
rng = np.random.default_rng(1)
users = o.synth_rows(o.SEED_QUERY, 0, 3, d)
sizes = [5000, 1, 333]
cands = [rng.integers(0, n, s).astype(np.uint32) for s in sizes]
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
for prec in (0, 1):
    m = pa.RankModel(ctx, pa.MODEL_DNN3, prec, blob)
    got = m.rank_dnn3(t, users, np.concatenate(cands), off)
    ref = np.concatenate([o.dnn3_forward(w, prec, users[r], tab[cands[r]]) for r in range(3)])
    diff = np.abs(got.astype(np.float64) - ref)
    print(f"DNN3 prec={prec}: max|d|={diff.max():.3e} mean={diff.mean():.3e} frac>1e-5={np.mean(diff>1e-5):.4f} bitexact={np.array_equal(got.view(np.uint32), ref.view(np.uint32))} nbit_mismatch={np.sum(got.view(np.uint32)!=ref.view(np.uint32))}")
    print("   sample", got[:4], ref[:4])
    m.destroy()
# FM two-tower
fw = o.Fm2tWeights(vocab=5000)
fblob = pa.pack_fm2t(fw)
ufids = rng.integers(0, 5000, (3, 8)).astype(np.int32)
ifids = rng.integers(0, 5000, (int(off[-1]), 8)).astype(np.int32)
for prec in (0, 1):
    m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, prec, fblob)
    got = m.rank_fm2t(users, ufids, ifids, off)
    ref = np.concatenate([o.fm2t_forward(fw, prec, users[r], ufids[r], ifids[off[r]:off[r+1]]) for r in range(3)])
    diff = np.abs(got.astype(np.float64) - ref)
    print(f"FM2T prec={prec}: max|d|={diff.max():.3e} mean={diff.mean():.3e} frac>1e-5={np.mean(diff>1e-5):.4f} nbit_mismatch={np.sum(got.view(np.uint32)!=ref.view(np.uint32))}")
    print("   sample", got[:4], ref[:4])
    m.destroy()
# throughput: R requests x 5000 candidates, device resident
R = 64
nI = R * 5000
cand = rng.integers(0, n, nI).astype(np.uint32)
offs = (np.arange(R + 1) * 5000).astype(np.uint32)
us = o.synth_rows(o.SEED_QUERY, 0, R, d)
d_u, d_c, d_o = ctx.to_device(us), ctx.to_device(cand), ctx.to_device(offs)
d_out = ctx.malloc(nI * 4)
for prec in (1, 0):
    m = pa.RankModel(ctx, pa.MODEL_DNN3, prec, blob)
    for it in range(3):
        ctx.synchronize(); t0 = time.time()
        for _ in range(10):
            m.rank_dnn3_dev(t, d_u, d_c, d_o, R, nI, d_out)
        ctx.synchronize(); dt = (time.time() - t0) / 10
        print(f"DNN3 prec={prec} R={R}: {dt*1e3:.3f} ms/call -> {nI/dt/1e6:.1f} M items/s, {nI*524800/dt/1e12:.1f} TFLOP/s(alg)")
    m.destroy()
