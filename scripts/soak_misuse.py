"""Soak: the C ABI under misuse — zero / oversized K and batch sizes, empty inputs, candidate rows outside the table,
request offsets that do not ascend or overshoot, zero-sized re-rank calls — every call must return an error code (or a defined
empty answer), never crash or fault the GPU.  Each case runs, then a normal recall checks the context still works.
Usage: soak_misuse.py"""
import os, sys
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

ctx = pa.Context(0)
n, d = 20_000, 128
tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
t = pa.Table(ctx, n, d)
t.upload(tab)
w = o.Dnn3Weights()
m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
q = o.synth_rows(o.SEED_QUERY, 0, 4, d)
want = t.recall_topk(q, 10)[0].copy()
results = []


def case(name, f):
    try:
        r = f()
        out = "returned " + (str(getattr(r, "shape", None)) if r is not None else "None")
    except pa._lib.PgError as ex:
        out = "error %d" % ex.code
    except (ValueError, IndexError, TypeError) as ex:         # refused by the Python wrapper itself
        out = "wrapper: " + type(ex).__name__
    ok = np.array_equal(t.recall_topk(q, 10)[0], want)          # the context still answers
    results.append((name, out, ok))
    print(f"{name:58s} {out:24s} context ok: {ok}", flush=True)


big = np.arange(300 * d, dtype=np.float32).reshape(300, d)
case("recall k = 0", lambda: t.recall_topk(q, 0)[0])
case("recall k = 16385", lambda: t.recall_topk(q, 16385)[0])
case("recall k = 16384 (> rows)", lambda: t.recall_topk(q, 16384)[0])
case("recall 0 queries", lambda: t.recall_topk(q[:0], 10)[0])
case("recall 300 queries (wrapper splits)", lambda: t.recall_topk(big, 10)[0])
case("l2 recall k = 0", lambda: t.recall_topk_l2(q, 0)[0])
case("rank: candidate row = rows", lambda: m.rank_dnn3(t, q[:1], np.array([5, n], np.uint32), [0, 2]))
case("rank: candidate row = 2^32-1", lambda: m.rank_dnn3(t, q[:1], np.array([0xFFFFFFFF], np.uint32), [0, 1]))
case("rank: offsets descend", lambda: m.rank_dnn3(t, q[:2], np.arange(10, dtype=np.uint32), [0, 8, 4]))
case("rank: offsets overshoot the candidates", lambda: m.rank_dnn3(t, q[:1], np.arange(10, dtype=np.uint32), [0, 50]))
case("rank: offsets start above 0", lambda: m.rank_dnn3(t, q[:1], np.arange(10, dtype=np.uint32), [3, 10]))
case("rank: zero requests", lambda: m.rank_dnn3(t, q[:0], np.zeros(0, np.uint32), [0]))
case("sort: offsets descend", lambda: ctx.sort_scores(np.arange(10.0), np.array([0, 8, 4], np.uint32)))
case("sort: offsets overshoot", lambda: ctx.sort_scores(np.arange(10.0), np.array([0, 50], np.uint32)))
case("sort: empty", lambda: ctx.sort_scores(np.zeros(0), np.array([0], np.uint32)))
case("dpp: no candidates", lambda: pa.dpp(ctx, t, np.zeros(0, np.uint32), np.zeros(0), 1.0, 10, 10))
case("dpp: topn 0", lambda: pa.dpp(ctx, t, np.arange(5, dtype=np.uint32), np.ones(5), 1.0, 0, 10))
case("dpp: window 0", lambda: pa.dpp(ctx, t, np.arange(50, dtype=np.uint32), np.linspace(1, 0.5, 50), 1.0, 10, 0))
case("dpp: candidate row outside the table", lambda: pa.dpp(ctx, t, np.array([1, n + 7], np.uint32), np.ones(2), 1.0, 2, 10))
case("dpp: NaN relevance", lambda: pa.dpp(ctx, t, np.arange(20, dtype=np.uint32), np.full(20, np.nan), 1.0, 5, 10))
case("ssd: no candidates", lambda: pa.ssd(ctx, t, np.zeros(0, np.uint32), np.zeros(0), 0.25, 10, 5)[0])
case("ssd: candidate row outside the table", lambda: pa.ssd(ctx, t, np.array([1, n + 7], np.uint32), np.ones(2), 0.25, 2, 5)[0])
case("i2i: trigger row outside the table", lambda: t.i2i_recall(np.array([n], np.uint32), 10)[0])
case("gather: row outside the table", lambda: t.gather(np.array([n], np.uint32)))
case("upload past the end", lambda: t.upload(tab[:10], n - 5))
ex = pa.Expr("${a}/${b}")
case("expr: zero items", lambda: ex.eval(ctx, np.zeros((2, 0))))
case("recommend: k = 0", lambda: pa.recommend_dnn3(ctx, t, m, pa.Expr("${gpu_dnn}"), "gpu_dnn", q, 0)[0])
case("recommend: unknown rank variable", lambda: pa.recommend_dnn3(ctx, t, m, pa.Expr("${other}"), "gpu_dnn", q, 10)[0])
bad = [r for r in results if not r[2]]
print(f"soak_misuse: {len(results)} cases, {len(bad)} left the context unusable", flush=True)
sys.exit(1 if bad else 0)
