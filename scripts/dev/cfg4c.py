"""developer aid: where the materialised-record FM + two-tower kernel spends its time (catalogue size, cached candidates)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from pairec_amd import _lib
from oracle import oracle as o
ctx = pa.Context(0)
R, K, vocab = 256, 5000, 1_000_000
n = R * K
fw = o.Fm2tWeights(vocab=vocab)
m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, pa.PREC_BF16, pa.pack_fm2t(fw))
rng = np.random.default_rng(5)
users = o.synth_rows(o.SEED_QUERY, 0, R, 128)
ufids = rng.integers(0, vocab, (R, 8)).astype(np.int32)
off = (np.arange(R + 1) * K).astype(np.uint32)
d_u, d_uf, d_off = ctx.to_device(users), ctx.to_device(ufids), ctx.to_device(off)
d_out = ctx.malloc(n * 4)
def timeit(fn, steps=20):
    for _ in range(3): fn()
    ctx.synchronize()
    ms = []
    for _ in range(steps):
        fn(); ms.append(ctx.stats().last_rank_ms)
    return float(np.mean(ms))
for n_cat in (int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 1_000_000):
    feats = pa.Features(ctx, n_cat)
    cols = ["if%d" % f for f in range(8)]
    for c_ in cols:
        feats.set_column(c_, pa.F_I32, rng.integers(0, vocab, n_cat).astype(np.int32))
    ir = pa.ItemRows(m, feats, cols)
    for name, cand in (("random", rng.integers(0, n_cat, n).astype(np.uint32)), ("row0", np.zeros(n, np.uint32)),
                       ("sequential", (np.arange(n) % n_cat).astype(np.uint32))):
        d_c = ctx.to_device(cand)
        t = timeit(lambda: _lib.check(ctx.L.pg_rank_fm2t_irows_dev(ctx.h, m.h, ir.h, d_u, d_uf, d_c, d_off, R, n, d_out)))
        t2 = timeit(lambda: _lib.check(ctx.L.pg_rank_fm2t_rows_dev(ctx.h, m.h, feats.h, pa.engine._ptr(feats._cols(cols)), d_u, d_uf, d_c, d_off, R, n, d_out)))
        print("catalogue %9d  candidates %-10s  item records %.3f ms   feature columns + per-field %.3f ms" % (n_cat, name, t, t2), flush=True)
        ctx.free(d_c)
    ir.destroy(); feats.destroy()
