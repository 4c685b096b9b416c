"""Developer aid: the split-bf16 rank stage alone (dnn3_x3_kernel, 256 x 5000 random candidate rows of a 100 M-row table), ms per
call; PG_LIB_PATH selects the build (ablations compute wrong results by design: the check line says so).  GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from oracle import oracle as o

R, K = 256, 5000
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = pa.Context(0)
t = pa.Table(ctx, rows, 128)
t.fill_synthetic(o.SEED_TABLE)
rng = np.random.default_rng(5)
nI = R * K
cand = rng.integers(0, rows, nI).astype(np.uint32)
offs = (np.arange(R + 1) * K).astype(np.uint32)
us = o.synth_rows(o.SEED_QUERY, 0, R, 128)
d_u, d_c, d_o = ctx.to_device(us), ctx.to_device(cand), ctx.to_device(offs)
d_out = ctx.malloc(nI * 4)
w = o.Dnn3Weights()
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16X3, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
loop = len(sys.argv) > 2 and sys.argv[2] == "loop"       # (scripts/dev/power_probe.sh: ~8 s of back-to-back calls)
res = []
for _ in range(700 if loop else 5):
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m.rank_dnn3_dev(t, d_u, d_c, d_o, R, nI, d_out)
    ctx.synchronize()
    res.append((time.perf_counter() - t0) / 10 * 1e3)
out = np.empty(nI, dtype=np.float32)
ctx.d2h(out, d_out)
# (cheap fingerprint only: an ablation build shows a different one)
msg = " | scores finite %s, mean %.6f, first %s" % (bool(np.all(np.isfinite(out))), float(out.mean()), out[:3])
print("%s: x3 rank stage ms per call: %s (device %.3f)%s" % (os.environ.get("PG_LIB_PATH", "product"), " ".join("%.3f" % x for x in res),
                                                            ctx.stats().last_rank_ms, msg))
