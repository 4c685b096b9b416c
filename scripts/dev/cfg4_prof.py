"""developer aid: the item-record FM + two-tower rank alone (random candidates of a 20 M-item catalogue) — target of the
rocprofv3 passes in scripts/profile_cfg4_r4.sh"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from pairec_amd import _lib
from oracle import oracle as o
ctx = pa.Context(0)
R, K, vocab, n_cat = 256, 5000, 1_000_000, 20_000_000
n = R * K
fw = o.Fm2tWeights(vocab=vocab)
PREC = {"bf16": pa.PREC_BF16, "bf16x3": pa.PREC_BF16X3, "f32": pa.PREC_F32}[os.environ.get("CFG4_PREC", "bf16")]
m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, PREC, pa.pack_fm2t(fw))
rng = np.random.default_rng(5)
users = o.synth_rows(o.SEED_QUERY, 0, R, 128)
ufids = rng.integers(0, vocab, (R, 8)).astype(np.int32)
off = (np.arange(R + 1) * K).astype(np.uint32)
d_u, d_uf, d_off = ctx.to_device(users), ctx.to_device(ufids), ctx.to_device(off)
d_out = ctx.malloc(n * 4)
feats = pa.Features(ctx, n_cat)
cols = ["if%d" % f for f in range(8)]
for c_ in cols:
    feats.set_column(c_, pa.F_I32, rng.integers(0, vocab, n_cat).astype(np.int32))
ir = pa.ItemRows(m, feats, cols)
mode = sys.argv[1] if len(sys.argv) > 1 else "random"
cand = {"random": rng.integers(0, n_cat, n), "row0": np.zeros(n), "sequential": np.arange(n) % n_cat}[mode].astype(np.uint32)
d_c = ctx.to_device(cand)
for _ in range(12):
    _lib.check(ctx.L.pg_rank_fm2t_irows_dev(ctx.h, m.h, ir.h, d_u, d_uf, d_c, d_off, R, n, d_out))
ctx.synchronize()
print("device ms", ctx.stats().last_rank_ms)
if len(sys.argv) > 2 and sys.argv[2] == "sync":          # one call at a time, as bench.py's cfg-4 leg times it
    ms = []
    for _ in range(20):
        _lib.check(ctx.L.pg_rank_fm2t_irows_dev(ctx.h, m.h, ir.h, d_u, d_uf, d_c, d_off, R, n, d_out))
        ms.append(ctx.stats().last_rank_ms)
    print("device ms, one call at a time: mean %.4f min %.4f max %.4f" % (np.mean(ms), np.min(ms), np.max(ms)))
if len(sys.argv) > 2 and sys.argv[2] == "loop":          # ~6 s of back-to-back calls (scripts/dev/power_probe.sh samples rocm-smi beside it)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 9.0:
        for _ in range(64):
            _lib.check(ctx.L.pg_rank_fm2t_irows_dev(ctx.h, m.h, ir.h, d_u, d_uf, d_c, d_off, R, n, d_out))
        ctx.synchronize()
        k += 64
    print("loop: %.4f ms per call" % ((time.perf_counter() - t0) / k * 1e3))
