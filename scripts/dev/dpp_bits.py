"""developer aid: the DPP kernel matrix from both pipes against each other and the oracle, mismatch counts per case"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pairec_amd as pa
from pairec_amd import _lib
from oracle import oracle as o
ctx = pa.Context(0)
rng = np.random.default_rng(21)
for R, n, d, norm, spread in [(1, 64, 128, True, 0), (1, 64, 127, True, 0), (1, 64, 128, False, 0), (1, 64, 15, False, 0), (1, 64, 3, False, 0), (1, 500, 128, True, 12)]:
    emb = rng.standard_normal((R, n, d)).astype(np.float32)
    if spread:
        emb *= np.exp2(rng.integers(-spread, spread + 1, (R, n, d))).astype(np.float32)
    rel = np.sort(rng.random((R, n)), axis=1)[:, ::-1].copy()
    d_e, d_r, d_L = ctx.to_device(emb), ctx.to_device(rel), ctx.malloc(R * n * n * 8)
    F = o.dpp_features(emb[0], None, norm, True)
    want = o.dpp_kernel_matrix_f(F, rel[0], 0.7)
    got = {}
    for valu in (0, 1):
        ctx.set_option("dpp_valu", valu)
        L = np.zeros((R, n, n))
        _lib.check(ctx.L.pg_dpp_kernel_matrix_dev(ctx.h, d_e, d_r, R, n, d, 0.7, int(norm), d_L))
        ctx.d2h(L, d_L)
        got[valu] = L[0]
    ne = lambda a, b: int((a.view(np.uint64) != b.view(np.uint64)).sum())
    bad = np.argwhere(got[0].view(np.uint64) != want.view(np.uint64))
    print((R, n, d, norm, spread), "mfma!=oracle", ne(got[0], want), "valu!=oracle", ne(got[1], want), "mfma!=valu", ne(got[0], got[1]),
          "first bad", bad[:4].tolist())
