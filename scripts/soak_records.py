"""Soak: the hit-record path of the > 128-query int8 screen (csrc/recall.hip) — and the threshold model feeding it — against
the exact fp32 scan on mid-size tables of several value distributions and row orders, random K and batch sizes 129..256:
rows and score bits must be identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = pa.Context(0)
ctx.set_option("predict_min_rows", "0")
rng = np.random.default_rng(19)
n, d = 5_000_000, 128
t = pa.Table(ctx, n, d)
bad = total = 0
t_end = time.time() + seconds
kinds = ("uniform", "gauss", "hot_head", "gauss_scaled_rows", "sorted_by_norm")
while time.time() < t_end:
    for kind in kinds:
        if kind == "uniform":
            t.fill_synthetic(o.SEED_TABLE + total)
        else:
            t.fill_gaussian(5 + total, 0.3)
            if kind != "gauss":
                chunk = 500_000
                for r0 in range(0, n, chunk):
                    rows = t.download(r0, chunk)
                    if kind == "hot_head":
                        if r0 == 0:
                            rows[:20_000] *= 2.5
                    elif kind == "gauss_scaled_rows":
                        rows *= rng.uniform(0.5, 1.5, (chunk, 1)).astype(np.float32)
                    else:
                        rows *= np.float32(1.0 + 1.5 * (1.0 - r0 / n))          # norms fall with the row index
                    t.upload(rows, r0)
        eb = t.screen_info()[0]
        k_fixed = int(rng.choice([50, 700, 5000]))
        for it in range(8):
            # (a fixed K for most batches of a table, so that the threshold model takes over after 1024 queries)
            k = k_fixed if it else int(rng.integers(1, 9000))
            nq = int(rng.integers(129, 257))
            q = rng.standard_normal((nq, d)).astype(np.float32) if it % 2 else o.synth_rows(o.SEED_QUERY, int(rng.integers(0, 900)), nq, d)
            rows, sc, cnt = t.recall_topk(q, k)
            sel = rng.choice(nq, 24, replace=False)
            total += 1
            # reference: the library's exact fp32 scan of the same table (no screen, no threshold model), a random subset
            ctx.set_option("recall_exact", "1")
            d_rows, d_sc, _ = t.recall_topk(q[sel], k)
            ctx.set_option("recall_exact", "0")
            if not (np.array_equal(rows[sel], d_rows) and np.array_equal(sc[sel].view(np.uint32), d_sc.view(np.uint32))):
                bad += 1
                print("MISMATCH", kind, "k", k, "nq", nq, "shadow", eb, flush=True)
        st = ctx.stats()
        print(f"{kind}: shadow elem bytes {eb}, batches {total}, mismatches {bad}, predicted {st.recall_predicted}, rescans {st.recall_rescans}", flush=True)
print("soak_records:", total, "batches,", bad, "mismatches")
sys.exit(1 if bad else 0)
