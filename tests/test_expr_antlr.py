"""ASTType "antlr" (GetExpASTWithType / ExprASTResultWithType, utils/ast/ast.go:338-389): the subset of the go-antlr-valuate
language that the reference's own tests pin (utils/ast/ast_test.go:30-56,90-167,213-300; functions
utils/ast/antlr_functions.go:34-91) — compiled to the same device program as the default grammar, everything else refused
by name.  CPU: the oracle's restatement against the transcribed vectors, the compiler's variables and refusals through the
C ABI (host code, no GPU).  GPU: the device evaluation against the vectors and the oracle, the mirror end to end."""
import json
import math

import numpy as np
import pytest

import pairec_amd as pa
from oracle import oracle as o
from test_host_mirror import H, CONFIG  # noqa: F401 — the mirror library fixture


def _expected(c):
    if "expect" in c:
        return c["expect"]
    if "expect_formula" in c:
        a, b, cc = (c["data"][n] for n in c["formula_args"])
        return (a + 2 * b) * math.pow(cc, 0.1)                  # the reference asserts r == this on its own machine
    it = o.OracleItem("x")                                      # "both ASTTypes give DeepEqual results"
    for k, v in c["data"].items():
        it.add_algo_score(k, v)
    return o.expr_eval(o.expr_parse(c["expr"]), it.float_expr_data)


def test_oracle_antlr_subset_against_the_reference_vectors(golden):
    for c in golden["expr_antlr"]:
        assert o.antlr_result(o.antlr_parse(c["expr"]), c["data"]) == _expected(c), c["ref"]
    # ExprASTResultByAntlr: an evaluation error (a variable the data map lacks, a non-number) → 0
    ast = o.antlr_parse("${ctr} + ${nobody}")
    assert o.antlr_result(ast, {"ctr": 0.5}) == 0.0
    assert o.antlr_result(o.antlr_parse("${ctr} * 2"), {"ctr": "text"}) == 0.0
    assert o.antlr_result(o.antlr_parse("maxValue(${v})"), {"v": []}) == 0.0
    assert o.antlr_result(o.antlr_parse(""), {}) == 0.0
    # precedence and the float division of the subset
    assert o.antlr_result(o.antlr_parse("2+3*2^3/4-1"), {}) == 2 + 3 * 8 / 4 - 1
    assert o.antlr_result(o.antlr_parse("-(${a}^2)"), {"a": 3.0}) == -9.0
    assert o.antlr_result(o.antlr_parse("(-${a})^2"), {"a": 3.0}) == 9.0
    assert o.antlr_result(o.antlr_parse("2*-3"), {}) == -6.0 and o.antlr_result(o.antlr_parse("2^-1"), {}) == 0.5
    assert o.antlr_result(o.antlr_parse("1/${z}"), {"z": 0.0}) == math.inf
    for bad in ("-${a}^2", "${a} % 2", "log(${a})", "${a} > 1", "2 ** 3", "${a}^2^3", "'x'", "${a} ? 1 : 2", "hash(${a})"):
        with pytest.raises(o.AntlrUnsupported):
            o.antlr_parse(bad)


def test_compiler_serves_the_subset_and_refuses_the_rest_by_name():
    e = pa.Expr("(${cdn_probs_ctr}+2*${cdn_probs_cvr})*${log_price}^0.1", "antlr")
    assert e.var_names == ["cdn_probs_ctr", "cdn_probs_cvr", "log_price"]
    e.free()
    e = pa.Expr("maxIndex(${p}) + maxValue( ${p} ) * -${w}", "antlr")
    assert e.var_names == ["maxIndex(p)", "maxValue(p)", "w"]
    e.free()
    pa.Expr("", "antlr").free()                                 # GetExpASTByAntlr(""): nil
    for bad, word in (("${a} % 2", "'%'"), ("log(${a})", '"log"'), ("${a} > 1", "'>'"), ("2 ** 3", "'**'"),
                      ("${a}^2^3", "associativity"), ("${a} ? 1 : 2", "'?'"), ("maxIndex(1)", "${name}"), ("(${a}", "')'"),
                      ("hash32(${a})", '"hash32"')):
        with pytest.raises(pa._lib.PgError) as ei:
            pa.Expr(bad, "antlr")
        assert ei.value.code == -4 and word in str(ei.value) and "subset" in str(ei.value), (bad, str(ei.value))
    # any other ASTType is the default grammar (ast.go:338-343): '#' and '%' belong to it
    e = pa.Expr("${a} # ${b} % 3", "default")
    assert e.var_names == ["a", "b"]
    e.free()


@pytest.mark.gpu
def test_antlr_subset_on_the_device(ctx, golden):
    for c in golden["expr_antlr"]:
        e = pa.Expr(c["expr"], "antlr")
        vals = []
        for name in e.var_names:                                # the host fills list functions per item
            if name.startswith("maxIndex(") or name.startswith("maxValue("):
                lst = c["data"][name[9:-1]]
                vals.append(float(np.argmax(lst)) if name.startswith("maxIndex") else float(np.max(lst)))
            else:
                vals.append(c["data"][name])
        got = e.eval(ctx, np.array(vals, dtype=np.float64).reshape(-1, 1))[0]
        want = _expected(c)
        if "expect" in c:
            assert got == want, c["ref"]
        else:                                                   # ^ = pow: the device's within 2 ulp of libm's
            assert abs(got - want) <= 4e-16 * abs(want), c["ref"]
        e.free()
    # random expressions of the subset against the oracle's restatement, 2 000 items
    rng = np.random.default_rng(8)
    src = "(${a}+2*${b})*${c}^0.1 - ${a}/(${b}-${b}) * 0 + -(${c}^2)/3"
    e = pa.Expr(src, "antlr")
    v = rng.uniform(0.01, 2.0, (3, 2000))
    got = e.eval(ctx, v)                                        # x/0 is ±Inf here (no arithmetic error), Inf * 0 = NaN
    ast = o.antlr_parse(src)
    for i in range(0, 2000, 97):
        want = o.antlr_result(ast, {"a": v[0, i], "b": v[1, i], "c": v[2, i]})
        assert (math.isnan(got[i]) and math.isnan(want)) or abs(got[i] - want) <= 1e-15 * abs(want)
    e.free()


@pytest.mark.gpu
def test_mirror_serves_an_antlr_scene_and_refuses_what_it_cannot(H):
    import copy
    cfg = copy.deepcopy(CONFIG)
    cfg["RankConf"]["home_feed"]["ASTType"] = "antlr"
    cfg["RankConf"]["home_feed"]["RankScore"] = "log(${gpu_dnn})"
    assert not H.ph_engine_create(json.dumps(cfg).encode())
    msg = H.ph_last_error()
    assert b'ASTType "antlr"' in msg and b"RankConf[home_feed].RankScore" in msg and b'"log"' in msg
    # current_score is not in ExprData(): an antlr RankScore over it fails Evaluate for every item → Score 0 (ast.go:374-383)
    expr_src = "${gpu_dnn}*(1+${recall_hint})^0.1"
    cfg["RankConf"]["home_feed"]["RankScore"] = expr_src
    h = H.ph_engine_create(json.dumps(cfg).encode())
    assert h, H.ph_last_error()
    n, d = 20000, 128
    w = o.Dnn3Weights()
    user = o.synth_rows(o.SEED_QUERY, 3, 1, d)[0]
    vec = " ".join("%d:%s" % (i + 1, repr(float(v))) for i, v in enumerate(user))
    H.ph_set_user_vector(h, b"u1", vec.encode())
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    assert H.ph_engine_load_dnn3(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    out = json.loads(H.ph_recommend(h, b"u1", 20, b"home_feed"))
    assert len(out["items"]) == 20 and all(x["score"] == 0.0 for x in out["items"])      # recall_hint is nowhere: every item fails
    H.ph_engine_destroy(h)
    # with the variable supplied by the experiment's parameters the scene ranks: Score = gpu_dnn * (1 + 0.5)^0.1
    cfg["RankConf"]["home_feed"]["RankScore"] = "${gpu_dnn}*(1+0.5)^0.1"
    h = H.ph_engine_create(json.dumps(cfg).encode())
    assert h, H.ph_last_error()
    H.ph_set_user_vector(h, b"u1", vec.encode())
    assert H.ph_engine_load_dnn3(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    out = json.loads(H.ph_recommend(h, b"u1", 20, b"home_feed"))
    for x in out["items"]:
        assert abs(x["score"] - x["algo_scores"]["gpu_dnn"] * math.pow(1.5, 0.1)) <= 1e-15
    assert all(a["score"] >= b["score"] for a, b in zip(out["items"], out["items"][1:]))
    H.ph_engine_destroy(h)
