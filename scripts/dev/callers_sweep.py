"""Developer aid (GPU box): p50 / throughput of per-request callers through the coalescer, by caller count, with the rejoin hold
on and off.  python scripts/dev/callers_sweep.py [rows]"""
import ctypes as C, json, os, sys
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import pairec_amd as pa
from oracle import oracle as o
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = pa.Context(0)
t = pa.Table(ctx, rows, 128)
t.fill_synthetic(o.SEED_TABLE)
w = o.Dnn3Weights()
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16X3, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
ex = pa.Expr(bench.RANK_EXPR)
users = np.ascontiguousarray(o.synth_rows(o.SEED_QUERY, 0, 1000, 128))
spec = bench.LoadgenSpec(mode=0, user_vecs=users.ctypes.data, n_users=1000, dim=128, k=5000, top_n=100)
for rejoin in [int(x) for x in os.environ.get("PG_CS_REJOIN", "1,0").split(",")]:
    ctx.set_option("coalescer_rejoin", rejoin)
    co = pa.Coalescer(ctx, t, 5000, m, ex, "gpu_dnn", max_top_n=100, depth=3)
    bench.loadgen(co, spec, 768, 3.0, 2, 2, 5000)          # (trains the table's threshold model)
    for callers in [int(x) for x in os.environ.get("PG_CS_CALLERS", "1,4,8,16,32,64,128,256,768").split(",")]:
        r = bench.loadgen(co, spec, callers, 1.5, 2, 2, 5000)
        print("rejoin %d callers %4d: p50 %.2f ms p99 %.2f  avg batch %.1f  %.1f M items/s" % (rejoin, callers, r["p50_ms"], r["p99_ms"], r["avg_batch"], r["value"] / 1e6), flush=True)
    co.destroy()
