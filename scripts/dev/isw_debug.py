"""developer aid: intermediates of fm2t_isw_kernel (built with -DPG_ISW_DEBUG) against numpy on one request of 32 items"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
os.environ["PG_ISW_DEBUG_MODE"] = str(mode)
import pairec_amd as pa
from pairec_amd import _lib
from oracle import oracle as o
ctx = pa.Context(0)
R, K, vocab, n_cat = 1, 32, 3000, 500
fw = o.Fm2tWeights(vocab=vocab)
m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, pa.PREC_BF16, pa.pack_fm2t(fw))
rng = np.random.default_rng(5)
users = o.synth_rows(o.SEED_QUERY, 0, R, 128)
ufids = rng.integers(0, vocab, (R, 8)).astype(np.int32)
ids = rng.integers(0, vocab, (n_cat, 8)).astype(np.int32)
feats = pa.Features(ctx, n_cat)
cols = ["if%d" % f for f in range(8)]
for f, c_ in enumerate(cols):
    feats.set_column(c_, pa.F_I32, np.ascontiguousarray(ids[:, f]))
ir = pa.ItemRows(m, feats, cols)
cand = rng.integers(0, n_cat, R * K).astype(np.uint32)
off = (np.arange(R + 1) * K).astype(np.uint32)
d_u, d_uf, d_off, d_c = ctx.to_device(users), ctx.to_device(ufids), ctx.to_device(off), ctx.to_device(cand)
d_out = ctx.malloc(R * K * 4)
# the kernel is only taken for many items? call the device entry point directly
_lib.check(ctx.L.pg_rank_fm2t_irows_dev(ctx.h, m.h, ir.h, d_u, d_uf, d_c, d_off, R, R * K, d_out))
ctx.synchronize()
got = np.empty(R * K, np.float32)
ctx.d2h(got, d_out)
bf = o.f32_to_bf16_round
# numpy restatement of the pieces
emb = np.stack([np.concatenate([fw.field_emb[8 + f][ids[c, f]] for f in range(8)]) for c in cand])     # [32][128]
lin = np.stack([[fw.field_lin[8 + f][ids[c, f]] for f in range(8)] for c in cand])                      # [32][8]
su = sum(fw.field_emb[f][ufids[0, f]] for f in range(8))
qu = sum(fw.field_emb[f][ufids[0, f]] ** 2 for f in range(8))
linu = fw.fm_b + sum(fw.field_lin[f][ufids[0, f]] for f in range(8))
s = su + emb.reshape(32, 8, 16).sum(1)
q = qu + (emb.reshape(32, 8, 16) ** 2).sum(1)
fm = linu + lin.sum(1) + 0.5 * (s * s - q).sum(1)
X = bf(emb)
h1 = bf(np.maximum(X @ bf(fw.iw1) + fw.ib1, 0))
h2 = h1 @ bf(fw.iw2) + fw.ib2
uo = o.fm2t_user_embedding(fw, 1, users[0])
ref = {0: o.fm2t_forward(fw, 1, users[0], ufids[0], ids[cand]),
       1: fm, 2: X.reshape(32, 8, 2, 8)[:, :, 0, :].sum((1, 2)), 3: X.reshape(32, 8, 2, 8)[:, :, 1, :].sum((1, 2)),
       4: h2[:, :32].reshape(32, 4, 2, 4)[:, :, 0, :].sum((1, 2)), 5: h2[:, :32].reshape(32, 4, 2, 4)[:, :, 1, :].sum((1, 2)),
       6: fm + h2[:, :32] @ uo[:32], 7: h2[:, 32:] @ uo[32:]}[mode]
print("mode", mode, "max |got - ref|", np.abs(got - ref).max())
print(" got", got[:6])
print(" ref", np.asarray(ref)[:6])
