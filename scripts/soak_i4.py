"""Soak: the 4-bit small-batch screen against the int8 screen on mid-size tables of several value distributions,
random K and batch sizes 1..4 — rows and score bits must be identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

ctx = pa.Context(0)
rng = np.random.default_rng(11)
n, d = 6_000_000, 128
t = pa.Table(ctx, n, d)
bad = 0
total = 0
for kind in ("uniform", "gauss", "gauss_scaled_rows", "lognormal_rows", "sparse"):
    if kind == "uniform":
        t.fill_synthetic(o.SEED_TABLE)
    elif kind == "gauss":
        t.fill_gaussian(5, 0.3)
    else:
        t.fill_gaussian(9, 1.0)
        chunk = 500_000
        for r0 in range(0, n, chunk):
            rows = t.download(r0, chunk)
            if kind == "gauss_scaled_rows":
                rows *= rng.uniform(0.01, 3.0, (chunk, 1)).astype(np.float32)
            elif kind == "lognormal_rows":
                rows *= np.exp(rng.standard_normal((chunk, 1)) * 1.5).astype(np.float32)
            else:
                rows[rng.random((chunk, d)) < 0.9] = 0.0
            t.upload(rows, row0=r0)
    for trial in range(12):
        nq = int(rng.integers(1, 5))
        k = int(rng.choice([1, 2, 10, 200, 1000, 5000, 8192, 9000, 16384]))
        q = rng.standard_normal((nq, d)).astype(np.float32) * np.float32(rng.choice([1e-3, 1.0, 50.0]))
        res = []
        used = []
        for mode in ("1", "0"):
            ctx.set_option("no_screen_i4", mode)
            rows, sc, cnt = t.recall_topk(q, k)
            res.append((rows, sc))
            used.append(ctx.last_scan_kernel()[1])
        same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))
        total += 1
        if not same:
            bad += 1
            print("MISMATCH", kind, nq, k, flush=True)
        if trial == 0:
            print(f"{kind}: first trial nq={nq} k={k} bytes int8 {used[0]} / 4-bit {used[1]} same={same}", flush=True)
print(f"trials {total}, bad {bad}")
sys.exit(1 if bad else 0)
