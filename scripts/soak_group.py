"""Soak: the shard group (2..4 LOGICAL shards on one device: row-range shards, exchange of the per-shard top-K, identical merge,
owner-computes rank, fusion, sort) on STRUCTURED tables — best rows all in one shard, duplicate runs across shard boundaries
(ties broken by global row), zero rows and zero queries, K above a shard's row count — against the same pipeline on ONE table
(pg_recommend_dnn3_dev on another context): page ids and order, recall score bits, model scores (fp32 mode) and fused scores.
Usage: soak_group.py [seconds] [seed]"""
import os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = pa.Context(0)
w = o.Dnn3Weights()
blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
model = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, blob)
ex = pa.Expr("${gpu_dnn}*(1+${current_score})^0.1")
bits = lambda a: np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)
KINDS = ("plain", "ascending", "descending", "duplicate_runs", "zero_rows", "best_in_one_shard")
d = 128
t_end = time.time() + seconds
cases = bad = 0
while time.time() < t_end:
    kind = str(rng.choice(KINDS))
    shards = int(rng.integers(2, 5))
    n = int(rng.choice([5_003, 90_001, 400_000]))
    v = rng.standard_normal(d).astype(np.float32)
    v /= np.linalg.norm(v)
    noise = rng.standard_normal((n, d)).astype(np.float32)
    ramp = np.linspace(0.2, 1.0, n, dtype=np.float32)[:, None]
    if kind == "ascending":
        tab = v[None] * ramp + 0.002 * noise
    elif kind == "descending":
        tab = v[None] * ramp[::-1] + 0.002 * noise
    elif kind == "duplicate_runs":
        base = 0.1 * rng.standard_normal((max(n // 700, 4), d)).astype(np.float32)
        tab = np.repeat(base, 701, axis=0)[:n].copy()
        if tab.shape[0] < n:
            tab = np.concatenate([tab, 0.1 * noise[: n - tab.shape[0]]])
    elif kind == "zero_rows":
        tab = 0.1 * noise * (rng.random((n, 1)) < 0.3).astype(np.float32)
    elif kind == "best_in_one_shard":
        tab = 0.05 * noise
        s = int(rng.integers(0, shards))
        a, b = s * n // shards, (s + 1) * n // shards
        tab[a:b] += v[None] * rng.uniform(0.5, 1.0, (b - a, 1)).astype(np.float32)
    else:
        tab = 0.1 * noise
    tab = np.ascontiguousarray(tab, dtype=np.float32)
    del noise
    g = pa.ShardGroup([0] * shards)
    g.table_create(n, d)
    g.table_upload(tab)
    g.model_load(pa.MODEL_DNN3, pa.PREC_F32, blob)
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    for _ in range(3):
        R = int(rng.choice([1, 7, 64, 200]))
        k = int(rng.choice([10, 400, 3000]))
        k = min(k, n)
        top_n = int(min(k, rng.choice([1, 10, 50])))
        qk = rng.integers(0, 3)
        q = (v[None] + 0.05 * rng.standard_normal((R, d))).astype(np.float32) if qk == 0 else \
            (0.1 * rng.standard_normal((R, d)).astype(np.float32) if qk == 1 else tab[rng.integers(0, n, R)].copy())
        desc = dict(kind=kind, shards=shards, n=n, R=R, k=k, top_n=top_n, qk=int(qk))
        if os.environ.get("SOAK_TRACE"):
            print("case", desc, flush=True)
        try:
            rows, rec, rnk, fus, cnt = g.recommend(ex, "gpu_dnn", q, k, top_n)
            srow, ssc, srk, sfu, sord, scnt = pa.recommend_dnn3(ctx, t, model, ex, "gpu_dnn", q, k)
        except Exception as exn:
            print("FAILED CASE", desc, repr(exn), flush=True)
            bad += 1
            cases += 1
            continue
        ok = True
        for r in range(R):
            idx = sord[r, :top_n].astype(np.int64)
            ok = ok and cnt[r] == top_n and np.array_equal(rows[r], srow[r][idx]) and np.array_equal(bits(rec[r]), bits(ssc[r][idx])) \
                and np.array_equal(bits(rnk[r]), bits(srk[r][idx])) and np.array_equal(bits(fus[r]), bits(sfu[r][idx]))
        cases += 1
        if not ok:
            bad += 1
            print("MISMATCH", desc, flush=True)
    t.destroy()
    g.destroy()
print(f"soak_group: {cases} steps, {bad} bad", flush=True)
sys.exit(1 if bad else 0)
