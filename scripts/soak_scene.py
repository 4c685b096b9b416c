"""Soak: ONE scene coalescer serving every per-request flavour at once — vector recall, squared-Euclidean recall, i2i recall,
DNN rank of a request's candidates, DPPSort and SSDSort calls, the fused recommend call — from 96 threads issuing random
flavours in random bursts, every answer compared with the direct call made beforehand on another context.  Looks for races
between the queues (slots, staging buffers, re-plans), not speed.
Usage: soak_scene.py [seconds] [seed]"""
import os, sys, time, threading, random
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
n, d, k, top_n = 1_500_000, 128, 300, 20
ctx = pa.Context(0)
ref = pa.Context(0)
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
w = o.Dnn3Weights()
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
ex = pa.Expr("${gpu_dnn}*(1+${current_score})^0.1")
NQ = 768
q = o.synth_rows(o.SEED_QUERY, 0, NQ, d) * rng.uniform(0.7, 1.3, (NQ, 1)).astype(np.float32)
bits = lambda a: np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)
# reference answers by the direct calls
R = {}
rows = np.zeros((NQ, k), np.uint64); sc = np.zeros((NQ, k), np.float32)
lrows = np.zeros((NQ, k), np.uint64); lsc = np.zeros((NQ, k), np.float32)
rnk = np.zeros((NQ, k), np.float32); page = np.zeros((NQ, top_n), np.uint64); pfus = np.zeros((NQ, top_n), np.float64)
for b in range(0, NQ, 256):
    r_, rec, rk, fus, order, _ = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q[b:b + 256], k)
    rows[b:b + 256], sc[b:b + 256], rnk[b:b + 256] = r_, rec, rk
    page[b:b + 256] = np.take_along_axis(r_, order[:, :top_n].astype(np.int64), axis=1)
    pfus[b:b + 256] = np.take_along_axis(fus, order[:, :top_n].astype(np.int64), axis=1)
for b in range(0, NQ, 128):
    lrows[b:b + 128], lsc[b:b + 128], _ = t.recall_topk_l2(q[b:b + 128], k)
trig = rng.integers(0, n, NQ).astype(np.uint32)
irows, isc, _ = t.i2i_recall(trig[:256], k)
# re-rank references: DPP / SSD on each query's first 120 recalled rows
C_ = 120
dpp_ref, ssd_ref = [], []
for i in range(128):
    cand = rows[i][:C_].astype(np.uint32)
    rel = np.sort(sc[i][:C_].astype(np.float64))[::-1].copy()
    dpp_ref.append(pa.dpp_ex(ctx, t, cand, rel, 1.0, 30, 10)[0])
    ssd_ref.append(pa.ssd(ctx, t, cand, rel, 0.25, 30, 5)[0])
print("references ready", flush=True)
co = pa.Coalescer(ctx, t, k, expr=ex, algos=[("gpu_dnn", m)], max_top_n=top_n, max_wait_us=300, max_rerank_items=C_)
bad = []
count = [0] * 7
stop = time.time() + seconds
def worker(wid):
    r = random.Random(seed * 1000 + wid)
    while time.time() < stop:
        for _ in range(r.randint(1, 12)):
            f = r.randint(0, 6)
            i = r.randrange(NQ)
            try:
                if f == 0:
                    a, b_, c = co.recall(q[i]); ok = np.array_equal(a, rows[i]) and np.array_equal(bits(b_), bits(sc[i]))
                elif f == 1:
                    a, b_, c = co.recall_l2(q[i]); ok = np.array_equal(a, lrows[i]) and np.array_equal(bits(b_), bits(lsc[i]))
                elif f == 2:
                    i %= 256
                    a, b_, c = co.i2i_recall(int(trig[i])); ok = np.array_equal(a, irows[i]) and np.array_equal(bits(b_), bits(isc[i]))
                elif f == 3:
                    a = co.rank(0, q[i], rows[i].astype(np.uint32)); ok = np.array_equal(bits(a), bits(rnk[i]))
                elif f == 4:
                    i %= 128
                    a, _u = co.dpp(rows[i][:C_].astype(np.uint32), np.sort(sc[i][:C_].astype(np.float64))[::-1].copy(), 1.0, 30, 10)
                    ok = np.array_equal(a, dpp_ref[i])
                elif f == 5:
                    i %= 128
                    a, _u = co.ssd(rows[i][:C_].astype(np.uint32), np.sort(sc[i][:C_].astype(np.float64))[::-1].copy(), 0.25, 30, 5)
                    ok = np.array_equal(a, ssd_ref[i])
                else:
                    a, b_, c, fu, cnt = co.recommend(q[i], top_n)
                    ok = np.array_equal(a, page[i]) and np.array_equal(bits(fu), bits(pfus[i]))
                count[f] += 1
                if not ok:
                    bad.append((f, i))
            except Exception as exn:
                bad.append((f, i, repr(exn)))
        time.sleep(r.random() * 0.004)
th = [threading.Thread(target=worker, args=(i,)) for i in range(96)]
[x.start() for x in th]
[x.join() for x in th]
st = co.stats()
co.destroy()
print(f"soak_scene: requests per flavour (recall, l2, i2i, rank, dpp, ssd, recommend) {count}, bad {len(bad)} {bad[:5]}", flush=True)
sys.exit(1 if bad else 0)
