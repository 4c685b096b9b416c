"""Soak (round 5): (1) the ASTType "antlr" subset — random strings of the subset's grammar, mutated ones and noise: the
compiler (pg_expr_compile_typed) and the oracle's restatement (oracle.antlr_parse) must agree on accept / refuse and on the
variable list; accepted expressions evaluated on the device over random values (zeros, negatives, huge / tiny magnitudes)
must equal oracle.antlr_result up to pow's 2 ulp, incl. float division by zero (no arithmetic error in this evaluator) and
list functions.  (2) RankConfig.ScoreRewrite — random rewrite maps (sources that overwrite algorithm names, new names,
sources that do not compile) in front of a random RankScore through pg_recommend_dnn3 on a small table: fused scores equal
oracle.fuse_scores(…, score_rewrite=…) to 1e-12 relative.
Usage: soak_antlr_rewrite.py [seconds] [seed] [--no-gpu]"""
import math, os, sys, time
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
import pairec_amd as pa

args = [a for a in sys.argv[1:] if not a.startswith("--")]
seconds = float(args[0]) if args else 30.0
seed = int(args[1]) if len(args) > 1 else 1
use_gpu = "--no-gpu" not in sys.argv
rng = np.random.default_rng(seed)
ctx = pa.Context(0) if use_gpu else None
NAMES = ["ctr", "cvr", "price", "a_b", "x1"]
LISTS = ["probs", "cls_2"]


def gen(depth=0):
    r = rng.random()
    if depth > 4 or r < 0.3:
        k = rng.integers(0, 6)
        if k == 0:
            return "${%s}" % NAMES[int(rng.integers(0, len(NAMES)))]
        if k == 1:
            return str(int(rng.integers(0, 1000)))
        if k == 2:
            return "%.3f" % (rng.random() * 10)
        if k == 3:
            return "%de%d" % (int(rng.integers(1, 9)), int(rng.integers(0, 4)))
        if k == 4:
            return "%s(${%s})" % (("maxIndex", "maxValue")[int(rng.integers(0, 2))], LISTS[int(rng.integers(0, 2))])
        return "0"
    if r < 0.45:
        return "(" + gen(depth + 1) + ")"
    if r < 0.5:
        return "-" + gen(depth + 1)
    sp = " " if rng.random() < 0.3 else ""
    return gen(depth + 1) + sp + "+-*/^"[int(rng.integers(0, 5))] + sp + gen(depth + 1)


def mutate(s):
    k = rng.integers(0, 6)
    i = int(rng.integers(0, max(len(s), 1)))
    if k == 0:
        return s[:i] + s[i + 1:]
    if k == 1:
        return s[:i] + str(rng.choice(list("()+-*/^%#$ {}.e_x1\t\n?><'\"log"))) + s[i:]
    if k == 2:
        return s[:i]
    if k == 3:
        return s + str(rng.choice(list(")(+*$ \t^")))
    if k == 4:
        return s.replace("${", "$", 1)
    return s.replace("}", "", 1)


def values(n):
    v = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)
    v[rng.random(n) < 0.15] = 0.0
    ints = rng.random(n) < 0.1
    v[ints] = rng.integers(-3, 4, int(ints.sum()))             # a few small integers
    return v


t_end = time.time() + seconds
n_str = n_acc = n_eval = n_rw = bad = n_pow = 0
tab = w = t = m = None
if use_gpu:
    nrow = 20000
    t = pa.Table(ctx, nrow, 128)
    t.fill_synthetic(o.SEED_TABLE)
    w = o.Dnn3Weights()
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
while time.time() < t_end:
    # ---- (1) the antlr subset
    src = gen()
    if rng.random() < 0.4:
        src = mutate(src)
    if rng.random() < 0.05:
        src = "".join(str(rng.choice(list("()+-*/^%${}0123456789._e maxIndexVlu\t"))) for _ in range(int(rng.integers(0, 24))))
    n_str += 1
    try:
        ast = o.antlr_parse(src)
        o_ok = True
    except o.AntlrUnsupported:
        o_ok, ast = False, None
    except RecursionError:
        continue
    try:
        e = pa.Expr(src, "antlr")
        g_ok = True
    except pa._lib.PgError as ex:
        g_ok, e = False, None
        if ex.code != -4:
            bad += 1
            print("REFUSAL CODE", repr(src), ex, flush=True)
    if o_ok != g_ok:
        # (the compiler bounds program size and nesting; the oracle does not)
        if not (o_ok and not g_ok and ("too large" in str(ex) or "nesting" in str(ex))):
            bad += 1
            print("ACCEPT MISMATCH", repr(src), "oracle", o_ok, "compiler", g_ok, flush=True)
        continue
    if not g_ok:
        continue
    n_acc += 1
    known = all((nm[9:-1] in LISTS) if (nm.startswith("maxIndex(") or nm.startswith("maxValue(")) else (nm in NAMES) for nm in e.var_names)
    if use_gpu and ast is not None and known:               # (a mutated name the data lacks: the host's "missing → 0" rule, not the device's)
        n = 64
        data = {nm: values(n) for nm in NAMES}
        lists = {nm: rng.standard_normal((n, int(rng.integers(1, 6)))) for nm in LISTS}
        cols = []
        for name in e.var_names:
            if name.startswith("maxIndex(") or name.startswith("maxValue("):
                L = lists[name[9:-1]]
                cols.append(np.argmax(L, axis=1).astype(np.float64) if name.startswith("maxIndex") else np.max(L, axis=1))
            else:
                cols.append(data[name])
        got = e.eval(ctx, np.array(cols, dtype=np.float64)) if cols else e.eval(ctx, np.zeros((0, n)))
        for i in range(0, n, 7):
            dd = {nm: float(data[nm][i]) for nm in NAMES}
            dd.update({nm: lists[nm][i].tolist() for nm in LISTS})
            want = o.antlr_result(ast, dd)
            gi = float(got[i])
            # (a ^ whose base carries the 2 ulp of an earlier fractional power and whose exponent is in the hundreds multiplies
            #  that error by the exponent: expressions with ^ are held to 1e-10, the others to a few ulps)
            tol = 1e-10 if "^" in src else 4e-15
            okv = (math.isnan(gi) and math.isnan(want)) or gi == want or (math.isfinite(want) and abs(gi - want) <= tol * max(abs(want), 1e-300))
            n_eval += 1
            if not okv and "^" in src and o.pow_last_ulp_explains(lambda: o.antlr_result(ast, dd), gi):
                n_pow += 1                               # (a power inside an exponent of a negative base, …: oracle.pow_last_ulp_explains)
                continue
            if not okv:
                bad += 1
                print("VALUE MISMATCH", repr(src), dd, "device", gi, "oracle", want, flush=True)
                break
    if e is not None:
        e.free()
    # ---- (2) ScoreRewrite in front of a RankScore, through the device pipeline
    if use_gpu and rng.random() < 0.15:
        pool = ["gpu_dnn", "boost", "mix", "z9"]
        srcs = list(rng.choice(pool, int(rng.integers(1, 4)), replace=False))
        def small():
            atoms = ["${gpu_dnn}", "${current_score}", "0.5", "2", "${gpu_dnn}*${gpu_dnn}", "(1+${current_score})"]
            a, b = atoms[int(rng.integers(0, len(atoms)))], atoms[int(rng.integers(0, len(atoms)))]
            return a + "+-*"[int(rng.integers(0, 3))] + b
        rew = {s_: (small() if rng.random() > 0.15 else "${gpu_dnn} @ 1") for s_ in srcs}
        rank_src = "+".join("${%s}" % s_ for s_ in set(srcs) | {"gpu_dnn"}) + "*(1+${current_score})^0.1"
        ex = pa.Expr(rank_src)
        ex.set_score_rewrites(rew)
        q = o.synth_rows(o.SEED_QUERY, int(rng.integers(0, 1000)), 2, 128)
        k = int(rng.choice([50, 300]))
        rows, rec, rnk, fus, order, _ = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q, k)
        for r_ in range(2):
            for i in range(0, k, 13):
                it = o.OracleItem(str(i), float(rec[r_][i]))
                it.add_algo_score("gpu_dnn", float(rnk[r_][i]))
                o.fuse_scores(rank_src, [it], score_rewrite=rew)
                n_rw += 1
                if abs(fus[r_][i] - it.score) > 1e-12 * max(abs(it.score), 1e-300):
                    bad += 1
                    print("REWRITE MISMATCH", rank_src, rew, "device", fus[r_][i], "oracle", it.score, flush=True)
                    break
        ex.free()
print(f"soak_antlr_rewrite: {n_str} strings, {n_acc} accepted by both sides, {n_eval} device values, {n_rw} rewritten fused scores, {bad} bad "
      f"({n_pow} values differ by what pow's last ulp does to a discontinuous use of it)", flush=True)
sys.exit(1 if bad else 0)
