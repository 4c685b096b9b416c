"""Soak: the fused recall → DNN rank → RankScore → sort pipeline (pg_recommend_dnn3_dev: caller-made batches, and the
coalescer's per-request calls from threads) on STRUCTURED tables (ascending / descending score order, clustered best rows,
duplicate runs, equal rows, zero rows and zero queries, huge rows) — against the same stages called one by one on another
context.  Exercises the pipeline's re-plan / patch paths and the coalescer's re-run of a batch whose plan failed.
Usage: soak_pipeline_adversarial.py [seconds] [seed]"""
import os, sys, time, threading
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = pa.Context(0)
w = o.Dnn3Weights()
model = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
ex = pa.Expr("${gpu_dnn}*(1+${current_score})^0.1")
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
KINDS = ("ascending", "descending", "best_at_head", "best_at_tail", "duplicate_runs", "all_equal", "zero_rows", "huge_rows", "plain", "crowded")
d = 128
t_end = time.time() + seconds
cases = bad = 0
while time.time() < t_end:
    kind = str(rng.choice(KINDS))
    n = int(rng.choice([60_000, 700_000, 2_000_000])) if kind != "crowded" else 3_000_000
    v = rng.standard_normal(d).astype(np.float32)
    v /= np.linalg.norm(v)
    noise = rng.standard_normal((n, d)).astype(np.float32)
    ramp = np.linspace(0.2, 1.0, n, dtype=np.float32)[:, None]
    if kind == "ascending":
        tab = v[None] * ramp + 0.002 * noise
    elif kind == "descending":
        tab = v[None] * ramp[::-1] + 0.002 * noise
    elif kind == "crowded":                                   # nearly collinear rows in random order: the screened lists overflow
        tab = v[None] * rng.uniform(0.2, 1.0, (n, 1)).astype(np.float32) + 0.002 * noise
    elif kind in ("best_at_head", "best_at_tail"):
        tab = 0.1 * noise
        m = min(int(rng.integers(100, 30_000)), n // 4)
        a = 0 if kind == "best_at_head" else n - m
        tab[a:a + m] += v[None] * rng.uniform(0.8, 1.2, (m, 1)).astype(np.float32)
    elif kind == "duplicate_runs":
        base = rng.standard_normal((max(n // 5000, 4), d)).astype(np.float32) * np.float32(0.1)
        tab = np.repeat(base, 5000, axis=0)[:n].copy()
        if tab.shape[0] < n:
            tab = np.concatenate([tab, 0.1 * noise[: n - tab.shape[0]]])
    elif kind == "all_equal":
        tab = np.repeat(v[None] * np.float32(0.7), n, axis=0)
    elif kind == "zero_rows":
        tab = 0.1 * noise * (rng.random((n, 1)) < 0.3).astype(np.float32)
    elif kind == "huge_rows":
        tab = 0.1 * noise
        tab[rng.integers(0, n, 50)] *= np.float32(300.0)
    else:
        tab = 0.1 * noise
    tab = np.ascontiguousarray(tab, dtype=np.float32)
    n = tab.shape[0]
    del noise
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    for _ in range(3):
        R = int(rng.choice([1, 3, 64, 130, 256]))
        k = int(rng.choice([10, 300, 2000]))
        qk = rng.integers(0, 3)
        q = (v[None] + 0.05 * rng.standard_normal((R, d))).astype(np.float32) if qk == 0 else \
            (0.1 * rng.standard_normal((R, d)).astype(np.float32) if qk == 1 else tab[rng.integers(0, n, R)].copy())
        desc = dict(kind=kind, n=n, R=R, k=k, qk=int(qk))
        if os.environ.get("SOAK_TRACE"):
            print("case", desc, flush=True)
        try:
            rows, rec, rnk, fus, order, cnt = pa.recommend_dnn3(ctx, t, model, ex, "gpu_dnn", q, k)
            # the stages one by one
            srow, ssc, scnt = t.recall_topk(q, k)
            off = (np.arange(R + 1) * k).astype(np.uint32)
            cand = srow.reshape(-1).astype(np.uint32)
            srk = model.rank_dnn3(t, q, cand, off).reshape(R, k)
            sfu = ex.eval(ctx, np.stack([{"gpu_dnn": srk.reshape(-1).astype(np.float64), "current_score": ssc.reshape(-1).astype(np.float64)}[vn]
                                         for vn in ex.var_names])).reshape(R, k)
            sord = ctx.sort_scores(sfu.reshape(-1), off, descending=True).reshape(R, k)
            ok = (np.array_equal(rows, srow) and np.array_equal(bits(rec), bits(ssc)) and np.array_equal(bits(rnk), bits(srk)) and
                  np.array_equal(fus.view(np.uint64), sfu.view(np.uint64)) and np.array_equal(order, sord) and cnt.tolist() == scnt.tolist())
            # the coalescer's per-request calls (a page of 20) from R threads
            top_n = min(20, k)
            co = pa.Coalescer(ctx, t, k, model=model, expr=ex, rank_var="gpu_dnn", max_top_n=top_n, max_wait_us=1000)
            got = [None] * R
            th = [threading.Thread(target=lambda i=i: got.__setitem__(i, co.recommend(q[i], top_n))) for i in range(R)]
            [x.start() for x in th]
            [x.join() for x in th]
            co.destroy()
            for i in range(R):
                page = np.take_along_axis(srow[i], sord[i, :top_n].astype(np.int64), axis=0)
                pf = np.take_along_axis(sfu[i], sord[i, :top_n].astype(np.int64), axis=0)
                ok = ok and got[i] is not None and np.array_equal(got[i][0], page) and np.array_equal(got[i][3].view(np.uint64), pf.view(np.uint64))
        except Exception as exn:
            print("FAILED CASE", desc, repr(exn), flush=True)
            bad += 1
            cases += 1
            continue
        cases += 1
        if not ok:
            bad += 1
            print("MISMATCH", desc, flush=True)
    t.destroy()
print(f"soak_pipeline_adversarial: {cases} batches, {bad} bad", flush=True)
sys.exit(1 if bad else 0)
