"""Soak: random bursts of concurrent callers (recall / rank / recommend, random burst sizes and pacing) through one
coalescer, every answer compared with the direct single-request calls on another context.  Looks for races, not speed."""
import os, sys, threading, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n, d, k, top_n = 3_000_000, 128, 300, 40
ctx = pa.Context(0)
ref_ctx = pa.Context(0)
t = pa.Table(ctx, n, d)
t.fill_synthetic(o.SEED_TABLE)
w = o.Dnn3Weights()
blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, blob)
ex = pa.Expr("${gpu_dnn}*(1+${current_score})^0.1")
NQ = 4096
q = o.synth_rows(o.SEED_QUERY, 0, NQ, d)
# reference answers, 256 at a time, on the plain batch API
ref_rows = np.zeros((NQ, k), dtype=np.uint64)
ref_sc = np.zeros((NQ, k), dtype=np.float32)
ref_page = np.zeros((NQ, top_n), dtype=np.uint64)
ref_rank = np.zeros((NQ, k), dtype=np.float32)
for b in range(0, NQ, 256):
    rows, rec, rnk, fus, order, cnt = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q[b:b + 256], k)
    ref_rows[b:b + 256], ref_sc[b:b + 256], ref_rank[b:b + 256] = rows, rec, rnk
    ref_page[b:b + 256] = np.take_along_axis(rows, order[:, :top_n].astype(np.int64), axis=1)
print("reference ready", flush=True)
bad = []
done = [0]
lock = threading.Lock()
rng = random.Random(7)
t_end = time.time() + seconds
rounds = 0
while time.time() < t_end:
    depth = rng.choice([1, 2, 3])
    co = pa.Coalescer(ctx, t, k, m, ex, "gpu_dnn", max_top_n=top_n, max_rank_items=k, max_wait_us=rng.choice([0, 50, 500, 3000]),
                      depth=depth, max_batch=rng.choice([256, 64, 7]))
    for burst in range(6):
        nthreads = rng.choice([1, 2, 5, 33, 100, 300])
        picks = [rng.randrange(NQ) for _ in range(nthreads)]
        kinds = [rng.randrange(3) for _ in range(nthreads)]
        bar = threading.Barrier(nthreads)

        def work(i):
            qi, kind = picks[i], kinds[i]
            bar.wait()
            if rng.random() < 0.3:
                time.sleep(rng.random() * 0.002)
            try:
                if kind == 0:
                    rows, sc, cnt = co.recall(q[qi])
                    ok = np.array_equal(rows, ref_rows[qi]) and np.array_equal(sc.view(np.uint32), ref_sc[qi].view(np.uint32)) and cnt == k
                elif kind == 1:
                    out = co.rank_dnn3(q[qi], ref_rows[qi].astype(np.uint32))
                    ok = np.array_equal(out.view(np.uint32), ref_rank[qi].view(np.uint32))
                else:
                    r = co.recommend(q[qi], top_n)
                    ok = np.array_equal(r[0], ref_page[qi]) and r[4] == top_n
            except Exception as e_:            # noqa: BLE001
                ok = False
                with lock:
                    bad.append((qi, kind, repr(e_)))
            if not ok:
                with lock:
                    bad.append((qi, kind, "mismatch"))
            with lock:
                done[0] += 1
        th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join(60)
            if x.is_alive():
                print("HANG: a caller did not return", flush=True)
                os._exit(3)
    st = co.stats()
    co.destroy()
    rounds += 1
print(f"rounds {rounds}, requests {done[0]}, bad {len(bad)}", bad[:5])
sys.exit(1 if bad else 0)
