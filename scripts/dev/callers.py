"""developer aid: the concurrent-callers legs alone on a smaller table (what scripts/profile_r3.sh traces): 768 callers
through pg_coalescer_recommend, with and without the DPP stage"""
import ctypes as C, json, os, sys
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
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
ex = pa.Expr(bench.RANK_EXPR)
class A: pass
a = A(); a.dim = 128; a.page = 100; a.callers = 768; a.callers_seconds = 3.0
out = bench.concurrent_callers_leg(pa, o, ctx, t, m, ex, a, 5000)
print(json.dumps({k: out[k] for k in ("value", "p50_ms", "avg_batch")}), json.dumps({k: out["recommend_dpp"][k] for k in ("value", "p50_ms", "vs_solo", "vs_caller_made_batch")}))
