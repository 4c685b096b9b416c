"""cfg 4 shape through the host-buffer API (kernel times are read with scripts/kstats_py.sh):
256 requests x 5000 candidates, FM over 8 + 8 fields (k = 16) + two-tower (128 -> 256 -> 64)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pairec_amd as pa           # noqa: E402
from oracle import oracle as o    # noqa: E402

ctx = pa.Context(0)
fw = o.Fm2tWeights(vocab=1_000_000)
R, K = 256, 5000
rng = np.random.default_rng(0)
users = o.synth_rows(o.SEED_QUERY, 0, R, 128)
ufids = rng.integers(0, 1_000_000, (R, 8)).astype(np.int32)
ifids = rng.integers(0, 1_000_000, (R * K, 8)).astype(np.int32)
off = (np.arange(R + 1) * K).astype(np.uint32)
for prec in (pa.PREC_BF16, pa.PREC_F32):
    m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, prec, pa.pack_fm2t(fw))
    m.rank_fm2t(users, ufids, ifids, off)
    t0 = time.perf_counter()
    for _ in range(5):
        m.rank_fm2t(users, ufids, ifids, off)
    dt = (time.perf_counter() - t0) / 5
    print("prec %d: %.2f ms per call incl. host copies (%.1f M items/s)" % (prec, dt * 1e3, R * K / dt / 1e6))
    m.destroy()
