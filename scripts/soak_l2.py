"""Soak: squared-Euclidean recall — the int8-screened pass (where the rows' norms allow it) against the exact scan of the same
table (knob l2_exact) on several norm profiles, random K and batch sizes: rows and distance bits must be identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = pa.Context(0)
rng = np.random.default_rng(23)
n, d = 4_000_000, 128
t = pa.Table(ctx, n, d)
bad = total = screened = 0
t_end = time.time() + seconds
while time.time() < t_end:
    for kind in ("normalised", "jitter_2pct", "jitter_10pct", "gauss", "zero_rows"):
        t.fill_synthetic(o.SEED_TABLE + total)
        if kind != "normalised":
            chunk = 500_000
            for r0 in range(0, n, chunk):
                rows = t.download(r0, chunk)
                if kind == "jitter_2pct":
                    rows *= rng.uniform(0.98, 1.02, (chunk, 1)).astype(np.float32)
                elif kind == "jitter_10pct":
                    rows *= rng.uniform(0.9, 1.1, (chunk, 1)).astype(np.float32)
                elif kind == "gauss":
                    rows[:] = rng.standard_normal((chunk, d)).astype(np.float32) * 0.1
                else:
                    rows[rng.random(chunk) < 0.01] = 0.0
                t.upload(rows, r0)
        for it in range(6):
            k = int(rng.choice([1, 60, 900, 5000]))
            nq = int(rng.integers(1, 129))
            q = (rng.standard_normal((nq, d)) * rng.uniform(0.05, 2.0)).astype(np.float32)
            rows, dist, cnt = t.recall_topk_l2(q, k)
            screened += ctx.last_scan_kernel()[1] < n * d * 2
            ctx.set_option("l2_exact", "1")
            sel = rng.choice(nq, min(nq, 16), replace=False)
            erows, edist, _ = t.recall_topk_l2(q[sel], k)
            ctx.set_option("l2_exact", "0")
            total += 1
            if not (np.array_equal(rows[sel], erows) and np.array_equal(dist[sel].view(np.uint32), edist.view(np.uint32))):
                bad += 1
                print("MISMATCH", kind, "k", k, "nq", nq, flush=True)
        print(f"{kind}: batches {total}, on the screened pass {screened}, mismatches {bad}, rescans {ctx.stats().recall_rescans}", flush=True)
print("soak_l2:", total, "batches,", bad, "mismatches")
sys.exit(1 if bad else 0)
