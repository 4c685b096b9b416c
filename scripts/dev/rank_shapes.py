"""Developer probe: DNN3 rank rate per hidden shape (bf16), 256 requests x 5000 candidate rows of a 20 M x 128 table."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

ctx = pa.Context(0)
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 128
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
R, K = 256, 5000
nI = R * K
rng = np.random.default_rng(1)
cand = rng.integers(0, n, nI).astype(np.uint32)
offs = (np.arange(R + 1) * K).astype(np.uint32)
us = o.synth_rows(o.SEED_QUERY, 0, R, d)
d_u, d_c, d_o = ctx.to_device(us), ctx.to_device(cand), ctx.to_device(offs)
d_out = ctx.malloc(nI * 4)
for h1, h2 in ((128, 128), (256, 128), (256, 256), (512, 256), (1024, 512)):
    w = o.Dnn3Weights(d_user=128, d_item=128, h1=h1, h2=h2, seed=o.SEED_WEIGHTS ^ (h1 + h2))
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    alg = 2 * (256 * h1 + h1 * h2 + h2)
    for no_ws in ((0, 1) if (h1, h2) == (512, 256) else (0,)):
        ctx.set_option("rank_no_ws", str(no_ws))
        best = 1e9
        for it in range(4):
            ctx.synchronize(); t0 = time.time()
            for _ in range(20):
                m.rank_dnn3_dev(t, d_u, d_c, d_o, R, nI, d_out)
            ctx.synchronize(); best = min(best, (time.time() - t0) / 20)
        st = ctx.stats()
        print(f"DNN3 256-{h1}-{h2}-1 bf16{' (streaming kernel)' if no_ws else ''}: {best*1e3:.3f} ms per 1.28 M items (device {st.last_rank_ms:.3f}) "
              f"= {nI/best/1e9:.2f} G items/s, {nI*alg/best/1e12:.0f} TFLOP/s algorithmic = {nI*alg/best/2.5e15:.3f} of 2.5 PF", flush=True)
    ctx.set_option("rank_no_ws", "0")
    m.destroy()
