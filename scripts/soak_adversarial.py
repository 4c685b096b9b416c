"""Soak: recalls over STRUCTURED tables — score order ascending / descending in the row index, the best rows clustered at the
head / middle / tail, long runs of duplicates, equal rows, zero rows, a few huge rows, tiny values, a constant direction plus
noise — random dim (64..256), K (1..16384), batch size (1..256), both metrics; against the library's exact fp32 scan of the
same table on a random subset of the queries and against the oracle on three of them.  Looks for plan / capacity corner
cases (every row of a chunk a suspect, thresholds that never close, overflowing lists), not speed.
Usage: soak_adversarial.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = pa.Context(0)
if os.environ.get("SOAK_SMALL_TABLE_PLANS"):
    # the plans that only big tables take by default — the 4-bit screen of small batches, the threshold model, the refinement —
    # on these mid-size structured tables
    for opt, val in (("i4_min_rows", 1024), ("predict_min_rows", 0), ("refine_min_rows", 1024), ("i4m_max_pairs", 1e12), ("i4m_max_lambda", 1e6)):
        ctx.set_option(opt, val)
def bits(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return np.where(np.isnan(a), np.uint32(0x7FC00000), a.view(np.uint32))       # (a NaN's sign / payload is not part of the answer)
KINDS = ("ascending", "descending", "best_at_head", "best_in_middle", "best_at_tail", "duplicate_runs", "all_equal", "zero_rows",
         "huge_rows", "tiny_values", "direction_plus_noise", "two_clusters", "nonfinite_rows", "mixture")
t_end = time.time() + seconds
cases = bad = 0
while time.time() < t_end:
    kind = str(rng.choice(KINDS))
    d = int(rng.choice([64, 128, 128, 128, 192, 256]))
    n = int(rng.choice([1, 5, 31, 33, 100, 2_000, 40_000, 300_000, 1_200_000, 2_500_000]))
    v = rng.standard_normal(d).astype(np.float32)
    v /= np.linalg.norm(v)
    noise = rng.standard_normal((n, d)).astype(np.float32)
    ramp = np.linspace(0.2, 1.0, n, dtype=np.float32)[:, None]
    if kind == "ascending":
        tab = v[None] * ramp + 0.002 * noise
    elif kind == "descending":
        tab = v[None] * ramp[::-1] + 0.002 * noise
    elif kind in ("best_at_head", "best_in_middle", "best_at_tail"):
        tab = 0.1 * noise
        m = max(min(int(rng.integers(100, 30_000)), n // 4), 1)
        a = 0 if kind == "best_at_head" else (max(n // 2 - m, 0) if kind == "best_in_middle" else n - m)
        tab[a:a + m] += v[None] * rng.uniform(0.8, 1.2, (m, 1)).astype(np.float32)
    elif kind == "duplicate_runs":
        base = rng.standard_normal((max(n // 5000, 4), d)).astype(np.float32)
        tab = np.repeat(base, max(min(5000, n // 4), 1), axis=0)[:n].copy()
        if tab.shape[0] < n:
            tab = np.concatenate([tab, noise[: n - tab.shape[0]]])
    elif kind == "all_equal":
        tab = np.repeat(v[None] * np.float32(0.7), n, axis=0)
    elif kind == "zero_rows":
        tab = noise * (rng.random((n, 1)) < 0.3).astype(np.float32)
    elif kind == "huge_rows":
        tab = 0.3 * noise
        hi = rng.integers(0, n, min(50, n))
        tab[hi] *= np.float32(300.0)
    elif kind == "nonfinite_rows":                          # a few NaN / +-inf values in the table (such tables ride the exact scan)
        tab = 0.3 * noise
        for val in (np.nan, np.inf, -np.inf):
            tab[rng.integers(0, n, 3), rng.integers(0, d, 3)] = val
    elif kind == "tiny_values":
        tab = noise * np.float32(1e-6)
    elif kind == "mixture":                                 # clustered rows: a few centres, tight clusters, normalised
        cen = rng.standard_normal((int(rng.choice([2, 8, 40])), d)).astype(np.float32)
        cen /= np.linalg.norm(cen, axis=1, keepdims=True)
        tab = cen[rng.integers(0, cen.shape[0], n)] + np.float32(rng.choice([0.3, 0.1, 0.03])) / np.float32(np.sqrt(d)) * noise
        tab /= np.linalg.norm(tab, axis=1, keepdims=True)
    elif kind == "direction_plus_noise":
        tab = v[None] * np.float32(0.9) + 0.05 * noise
    else:
        w = rng.standard_normal(d).astype(np.float32)
        w /= np.linalg.norm(w)
        tab = np.where(rng.random((n, 1)) < 0.5, v[None], w[None]) + 0.02 * noise
    tab = np.ascontiguousarray(tab, dtype=np.float32)
    del noise
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    for _ in range(5):
        l2 = d in (64, 128) and rng.random() < 0.35
        nq = int(rng.choice([1, 2, 3, 4, 5, 7, 16, 32, 33, 48, 64, 65, 128, 129, 200, 256]))
        # (round 6: the refinement stage of crowded tables — int8 residual shadow, 16-bit queries — forced on for half the batches)
        ctx.set_option("r2_min_factor", -1 if rng.random() < 0.5 else 3)
        if d > 128:
            nq = min(nq, 32)
        if l2:
            nq = min(nq, 200)
        k = int(rng.choice([1, 3, 10, 100, 1000, 5000, 16384]))
        qk = rng.integers(0, 3)
        q = (v[None] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32) if qk == 0 else \
            (rng.standard_normal((nq, d)).astype(np.float32) if qk == 1 else tab[rng.integers(0, n, nq)].copy())
        if rng.random() < 0.15:                               # non-finite or enormous query components
            q[rng.integers(0, nq), rng.integers(0, d)] = rng.choice([np.nan, np.inf, -np.inf, 3e38, 1e-40])
        desc = dict(kind=kind, n=n, d=d, nq=nq, k=k, l2=bool(l2), qk=int(qk), q_finite=bool(np.all(np.isfinite(q))))
        if os.environ.get('SOAK_TRACE'):
            print('case', desc, flush=True)
        try:
            rows, sc, cnt = (t.recall_topk_l2 if l2 else t.recall_topk)(q, k)
        except Exception as ex:
            print("FAILED CASE", desc, ex, flush=True)
            bad += 1
            cases += 1
            continue
        m = min(k, n)
        sel = sorted(set(int(x) for x in rng.integers(0, nq, 6)))
        ctx.set_option("recall_exact", "1")
        er, es, _ = (t.recall_topk_l2 if l2 else t.recall_topk)(q[sel], k)
        ctx.set_option("recall_exact", "0")
        ok = np.array_equal(rows[sel], er) and np.array_equal(bits(sc[sel]), bits(es)) and cnt.tolist() == [m] * nq
        osel = sel[:2]
        orow, osc = (o.recall_topk_l2 if l2 else o.recall_topk)(tab, q[osel], k)
        ok = ok and np.array_equal(rows[osel][:, :m], orow[:, :m]) and np.array_equal(bits(sc[osel][:, :m]), bits(osc[:, :m]))
        cases += 1
        if not ok:
            bad += 1
            print("MISMATCH", desc, flush=True)
    t.destroy()
print(f"soak_adversarial: {cases} recalls, {bad} bad", flush=True)
sys.exit(1 if bad else 0)
