"""Developer probe: the FM + two-tower rank kernel with random, constant and sequential field ids — how much of its
time is the gather (random 64-B rows out of 8 x 64 MB tables) and how much the arithmetic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from pairec_amd import _lib
from oracle import oracle as o

ctx = pa.Context(0)
R, K, vocab = 256, 5000, 1_000_000
fw = o.Fm2tWeights(vocab=vocab)
m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, pa.PREC_BF16, pa.pack_fm2t(fw))
rng = np.random.default_rng(5)
users = o.synth_rows(o.SEED_QUERY, 0, R, 128)
ufids = rng.integers(0, vocab, (R, 8)).astype(np.int32)
n = R * K
off = (np.arange(R + 1) * K).astype(np.uint32)
d_u, d_uf, d_off = ctx.to_device(users), ctx.to_device(ufids), ctx.to_device(off)
d_out = ctx.malloc(n * 4)
for name, ids in (("random", rng.integers(0, vocab, (n, 8)).astype(np.int32)),
                  ("constant", np.zeros((n, 8), dtype=np.int32)),
                  ("sequential", (np.arange(n, dtype=np.int64)[:, None] % vocab + np.zeros((1, 8), dtype=np.int64)).astype(np.int32)),
                  ("random_64k", rng.integers(0, 65536, (n, 8)).astype(np.int32))):
    d_if = ctx.to_device(ids)
    ms = []
    for _ in range(8):
        _lib.check(ctx.L.pg_rank_fm2t_dev(ctx.h, m.h, d_u, d_uf, d_if, d_off, R, n, d_out))
        ms.append(ctx.stats().last_rank_ms)
    print(f"{name}: {np.mean(ms[3:]):.3f} ms", flush=True)
    ctx.free(d_if)
