"""CPU tests: the oracle against the reference's own known-answer tests (tests/golden/) and against
independent numpy restatements.  No GPU, no product code."""
import math
import sys
from fractions import Fraction

import numpy as np
import pytest

from oracle import oracle as o


# ---------------------------------------------------------------------------------------------
# reference-pinned: expression fusion, sort order, decoder widening
# ---------------------------------------------------------------------------------------------
def _item_from_case(c):
    it = o.OracleItem("item_1")
    for k, v in c["algo_scores"].items():
        it.add_algo_score(k, v)
    for k, v in c["properties"].items():
        it.add_property(k, v)
    return it


def test_expr_reference_known_answers(golden):
    for c in golden["expr"]:
        it = _item_from_case(c)
        got = o.expr_eval(o.expr_parse(c["expr"]), it.float_expr_data)
        if "expect" in c:
            assert got == c["expect"], c["ref"]
        else:
            vals = {**c["algo_scores"], **c["properties"]}
            a, b, cc = (vals[n] for n in c["formula_args"])
            # the reference asserts result == (a+2*b)*math.Pow(c,0.1) on its own machine
            assert got == (a + 2 * b) * math.pow(cc, 0.1), c["ref"]


def test_fuse_scores_mutates_item_score(golden):
    c = golden["expr"][0]
    it = _item_from_case(c)
    o.fuse_scores(c["expr"], [it])
    assert it.score == 0.5
    # AB params override when non-zero (service/rank/ast_parameter_data.go:30-40)
    it2 = _item_from_case(c)
    o.fuse_scores(c["expr"], [it2], {"ctr": 1.0})
    assert it2.score == 1.0 + 0.3 + 0.1


def test_current_score_side_effect():
    it = o.OracleItem("x", score=0.25)
    assert it.float_expr_data("current_score") == 0.25
    assert it.algo_scores["recall_score"] == 0.25      # module/item.go:190-197


def test_sort_reference_known_answers(golden):
    for c in golden["sort"]:
        got = o.sort_scores(c["scores"], c["descending"]).tolist()
        assert got == c["expect_order"], c["ref"]


def test_decode_reference_known_answers(golden):
    d = golden["decode"][0]
    arr = np.asarray(d["float_val"], dtype=np.float32).reshape(-1, d["dim1"])
    wide = o.widen_f32(arr)
    assert wide.dtype == np.float64
    assert np.float32(wide[d["item"], d["index"]]) == np.float32(d["expect_f32"])
    f = golden["decode"][1]
    assert o.alink_fm_score(f["label"], f["score"]) == 1 - f["score"]
    assert o.alink_fm_score(1.0, f["score"]) == f["score"]


# ---------------------------------------------------------------------------------------------
# expression language quirks (utils/ast/parse.go, ast.go)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("src,expect", [
    ("1+2*3", 7.0), ("(1+2)*3", 9.0), ("2^3^2", 64.0),       # '^' is left-associative here
    ("7%3", 1.0), ("0-7%3", -1.0), ("10/4", 2.5), ("0#4*2", 8.0), ("3#4", 3.0),
    ("-5", -5.0), ("2*1e-5", 0.0), ("1_000+1", 1001.0), ("1e3", 1000.0), ("2*", 2.0),
    ("1.5.2+1", 0.0), ("${missing}+1", 1.0), ("  4 + 4 ", 8.0),
])
def test_expr_quirks(src, expect):
    assert o.expr_eval(o.expr_parse(src), lambda n: None) == expect


def test_expr_errors():
    with pytest.raises(o.ExprError):
        o.expr_parse("abc")                         # symbol error
    with pytest.raises(o.ExprError):
        o.expr_parse("1 +\t")                       # stale trailing tab (parse.go:125-133)
    with pytest.raises(o.ExprError):
        o.expr_eval(o.expr_parse("1/0"), lambda n: None)
    with pytest.raises(o.ExprError):
        o.expr_eval(o.expr_parse("1%0"), lambda n: None)
    assert o.expr_parse("") is None


# ---------------------------------------------------------------------------------------------
# dedup, vector text formats
# ---------------------------------------------------------------------------------------------
def test_unique_filter_first_wins_and_merges():
    a = o.OracleItem("1", 0.5, "r1")
    b = o.OracleItem("2", 0.4, "r1")
    c = o.OracleItem("1", 0.9, "r2")
    c.add_algo_score("m", 0.7)
    out = o.unique_filter([a, b, c])
    assert [x.id for x in out] == ["1", "2"]
    assert out[0].score == 0.5 and out[0].recall_scores == {"r1": 0.5, "r2": 0.9}
    assert out[0].algo_scores["m"] == 0.7


def test_vector_string_and_cache_format():
    v = o.parse_vector_string("1:0.12 2:-0.3 junk 3:1e-2 4:x")
    assert v.dtype == np.float32 and v.tolist() == [np.float32(0.12), np.float32(-0.3), np.float32(0.01), 0.0]
    items = [o.OracleItem("a", 0.5), o.OracleItem("b", 1e-7), o.OracleItem("c", 3.0)]
    assert o.recall_cache_string(items, "vec") == "a:vec:0.5,b:vec:1e-07,c:vec:3"


# ---------------------------------------------------------------------------------------------
# numeric oracle vs independent numpy
# ---------------------------------------------------------------------------------------------
def test_synth_rows_definition():
    t = o.synth_rows(o.SEED_TABLE, 5, 4, 64, normalize=False)
    for r in range(4):
        for c in (0, 1, 63):
            u = o.splitmix64(o.SEED_TABLE ^ ((5 + r) * 64 + c))
            assert t[r, c] == np.float32((u >> 40) * 2.0 ** -23 - 1.0)
    tn = o.synth_rows(o.SEED_TABLE, 5, 4, 64, normalize=True)
    assert np.allclose(np.linalg.norm(tn.astype(np.float64), axis=1), 1.0, atol=1e-6)
    # any row is reproducible independently of the block it was generated in
    assert np.array_equal(o.synth_rows(o.SEED_TABLE, 7, 1, 64), tn[2:3])


def test_dot_scores_is_k_ordered_fma_chain():
    rng = np.random.default_rng(0)
    tab = rng.standard_normal((50, 64)).astype(np.float32)
    q = rng.standard_normal((3, 64)).astype(np.float32)
    got = o.dot_scores(tab, q)
    for qi in range(3):
        for r in (0, 17, 49):
            acc = np.float32(0)
            for k in range(64):   # fma in float64 is exact enough to emulate one fp32 rounding here
                acc = np.float32(np.float64(tab[r, k]) * np.float64(q[qi, k]) + np.float64(acc))
            assert got[qi, r] == acc


def test_recall_topk_order_and_ties():
    tab = np.zeros((300, 64), dtype=np.float32)
    tab[:, 0] = np.repeat(np.arange(100, dtype=np.float32), 3)      # triple ties
    q = np.zeros((1, 64), dtype=np.float32)
    q[0, 0] = 1.0
    rows, scores = o.recall_topk(tab, q, 7, row_offset=1000)
    assert rows[0].tolist() == [1297, 1298, 1299, 1294, 1295, 1296, 1291]   # score desc, row asc
    assert scores[0].tolist() == [99, 99, 99, 98, 98, 98, 97]
    # k > rows
    rows, scores = o.recall_topk(tab[:5], q, 9)
    assert rows.shape == (1, 5)


def test_recall_topk_l2_is_the_specified_chain_and_order():
    """Squared-Euclidean top-k (hologres_vector_recall_v2.go:23: ORDER BY distance ascending, Score = distance): the oracle's
    distances are fmaf(-2, ip, |x|^2 + |q|^2) with k-ascending fp32 fmaf chains — restated here with numpy scalars —, the
    order is distance ascending with ties by row, and it agrees with a float64 brute force wherever that one has no ties."""
    rng = np.random.default_rng(5)
    n, d, k = 700, 64, 40
    tab = (rng.standard_normal((n, d)) * rng.uniform(0.3, 2.0, (n, 1))).astype(np.float32)
    tab[50:54] = tab[50]                                        # ties: rows 50..53 in row order
    q = rng.standard_normal((3, d)).astype(np.float32)
    q[1] = tab[50]
    rows, dist = o.recall_topk_l2(tab, q, k)

    def chain(a, b):
        acc = np.float32(0.0)
        for x, y in zip(a, b):
            acc = np.float32(np.float64(x) * np.float64(y) + np.float64(acc))      # fmaf: one rounding of the exact product-sum
        return acc
    for qi in range(3):
        nq = chain(q[qi], q[qi])
        for j in (0, 1, 7, k - 1):
            r = int(rows[qi, j])
            ip, nx = chain(tab[r], q[qi]), chain(tab[r], tab[r])
            t = np.float32(nx + nq)
            want = np.float32(np.float64(-2.0) * np.float64(ip) + np.float64(t))
            assert dist[qi, j].view(np.uint32) == want.view(np.uint32)
        assert np.all(np.diff(dist[qi].astype(np.float64)) >= 0)
    assert rows[1, :4].tolist() == [50, 51, 52, 53]
    d64 = ((tab[None].astype(np.float64) - q[:, None].astype(np.float64)) ** 2).sum(-1)
    ref = np.argsort(d64, axis=1, kind="stable")[:, :k]
    assert np.array_equal(rows[0], ref[0]) and np.array_equal(rows[2], ref[2])


def test_topk_merge_equals_global():
    tab = o.synth_rows(o.SEED_TABLE, 0, 4000, 64)
    q = o.synth_rows(o.SEED_QUERY, 0, 1, 64)
    g_rows, g_scores = o.recall_topk(tab, q, 100)
    parts = [o.recall_topk(tab[s:s + 1000], q, 100, row_offset=s) for s in range(0, 4000, 1000)]
    m_rows, m_scores = o.topk_merge(np.stack([p[0][0] for p in parts]), np.stack([p[1][0] for p in parts]), 100)
    assert np.array_equal(m_rows, g_rows[0]) and np.array_equal(m_scores, g_scores[0])


def test_bf16_rounding_rne():
    x = np.array([1.0, 1.00390625, 1.005859375, -2.5, 3.0e38, 1e-40, np.inf], dtype=np.float32)
    r = o.f32_to_bf16_round(x)
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == np.float32(1.0078125)   # tie → even, above tie → up
    for v in x:
        assert o.lib().orc_bf16_to_f32(o.lib().orc_f32_to_bf16(float(v))) == o.f32_to_bf16_round(np.array([v]))[0]


def test_dnn3_matches_float64_numpy():
    w = o.Dnn3Weights(16, 128, 512, 256)
    user = o.synth_rows(o.SEED_QUERY, 0, 1, 16)[0]
    items = o.synth_rows(o.SEED_TABLE, 0, 20, 128)
    got = o.dnn3_forward(w, 0, user, items)
    x = np.concatenate([np.tile(user, (20, 1)), items], axis=1).astype(np.float64)
    h1 = np.maximum(x @ w.w1.astype(np.float64) + w.b1, 0)
    h2 = np.maximum(h1 @ w.w2.astype(np.float64) + w.b2, 0)
    ref = 1 / (1 + np.exp(-(h2 @ w.w3.astype(np.float64) + w.b3)))
    assert np.max(np.abs(got - ref)) < 2e-6
    got16 = o.dnn3_forward(w, 1, user, items)
    assert np.max(np.abs(got16 - ref)) < 5e-3 and np.max(np.abs(got16 - got)) > 0


def test_fm2t_matches_float64_numpy():
    w = o.Fm2tWeights(vocab=50)
    rng = np.random.default_rng(3)
    user = o.synth_rows(o.SEED_QUERY, 0, 1, 128)[0]
    uf = rng.integers(0, 50, 8)
    itf = rng.integers(0, 50, (10, 8))
    got = o.fm2t_forward(w, 0, user, uf, itf)
    f64 = np.float64
    u1 = np.maximum(user.astype(f64) @ w.uw1.astype(f64) + w.ub1, 0)
    uo = u1 @ w.uw2.astype(f64) + w.ub2
    ref = []
    for i in range(10):
        vs = [w.field_emb[f][uf[f]].astype(f64) for f in range(8)] + \
             [w.field_emb[8 + f][itf[i, f]].astype(f64) for f in range(8)]
        lin = w.fm_b + sum(f64(w.field_lin[f][uf[f]]) for f in range(8)) + \
            sum(f64(w.field_lin[8 + f][itf[i, f]]) for f in range(8))
        s = np.sum(vs, axis=0)
        cross = 0.5 * np.sum(s * s - np.sum([v * v for v in vs], axis=0))
        x = np.concatenate(vs[8:])
        io = np.maximum(x @ w.iw1.astype(f64) + w.ib1, 0) @ w.iw2.astype(f64) + w.ib2
        ref.append(1 / (1 + math.exp(-(lin + cross + float(uo @ io)))))
    assert np.max(np.abs(got - np.array(ref))) < 2e-6


def test_sort_ties_nan_and_signed_zero():
    s = [1.0, float("nan"), -0.0, 0.0, 1.0, -1.0]
    assert o.sort_scores(s, True).tolist() == [0, 4, 2, 3, 5, 1]
    assert o.sort_scores(s, False).tolist() == [5, 2, 3, 0, 4, 1]
    assert o.sort_scores([], True).tolist() == []


def _np_dpp(L, topn, window):
    """Independent restatement of DPPWithWindow/DPP (dpp_sort.go:477-551) in numpy."""
    def once(topn, existed):
        N = L.shape[0]
        topn = min(topn, N)
        d2 = np.array([np.nan if i in existed else L[i, i] for i in range(N)])
        def maxidx(v):
            best, ind = np.nan, 0
            for i, x in enumerate(v):
                if x != x:
                    continue
                if best != best or x > best:
                    best, ind = x, i
            return ind
        j = maxidx(d2)
        Y = [j]
        c = np.zeros((topn, N))
        while len(Y) < topn:
            dj = d2[j]
            if dj < 1e-10:
                break
            dj = math.sqrt(dj)
            k = len(Y) - 1
            e = L[j].copy()
            if k > 0:
                ss = np.zeros(N)
                for i in range(k):
                    ss = ss + c[i, j] * c[i]
                e = e - ss
            e = (1 / dj) * e
            c[k] = e
            d2 = d2 - e * e
            d2[j] = np.nan
            j = maxidx(d2)
            Y.append(j)
        if len(Y) < topn:
            for i in range(N):
                if i not in existed and i not in Y:
                    Y.append(i)
                    if len(Y) == topn:
                        break
        return Y
    res = []
    if topn <= window:
        return once(topn, res)
    for _ in range(topn // window):
        res = res + once(window, res)
    if topn % window:
        res = res + once(topn % window, res)
    return res


def test_dpp_matches_numpy_restatement():
    rng = np.random.default_rng(5)
    # clustered embeddings so diversity actually reorders things
    centers = rng.standard_normal((6, 32))
    emb = centers[rng.integers(0, 6, 80)] + 0.15 * rng.standard_normal((80, 32))
    emb = o.l2_normalize_f64(emb)
    rel = np.sort(rng.random(80))[::-1].copy()
    L = o.dpp_kernel_matrix(emb, rel, 1.0)
    F = np.concatenate([emb, np.ones((80, 1))], axis=1) * 0.70710678118654757
    r = np.exp(rel)
    Lref = (r[:, None] * (F @ F.T)) * r[None, :]
    assert np.allclose(L, Lref, rtol=1e-13, atol=0)
    for topn, window in ((10, 10), (25, 10), (7, 3), (80, 10)):
        got = o.dpp_with_window(L, topn, window).tolist()
        assert got == _np_dpp(L, topn, window), (topn, window)
        assert len(set(got)) == len(got)
    assert o.dpp_with_window(L, 25, 10).tolist() != list(range(25))    # diversity changed the order


def test_dpp_kernel_matrix_matches_exact_rational_restatement():
    """KernelMatrix (dpp_sort.go:407-472) restated with exact rationals for the fma chain of S = F F^T and Python
    floats for L = (r_i S_ij) r_j, over the four feature-row variants (table, table + hook, hook-only with and
    without EnsurePositiveSim); then the greedy picks of the oracle equal the numpy restatement on each L."""
    rng = np.random.default_rng(21)
    n, d, h = 24, 12, 5
    emb = (rng.standard_normal((n, d)) * 0.5).astype(np.float32)
    hook = rng.standard_normal((n, h))
    rel = np.sort(rng.random(n))[::-1].copy()

    def fma(x, y, s_):
        return float(Fraction(x) * Fraction(y) + Fraction(s_))
    for emb32, hk, norm, pos in ((emb, None, True, True), (emb, hook, True, True), (None, hook, True, True),
                                 (None, hook, False, False), (emb, None, False, True)):
        F = o.dpp_features(emb32, hk, norm, pos)
        w = (0 if emb32 is None else d) + (0 if hk is None else h)
        assert F.shape == (n, w + 1)
        if emb32 is not None or pos:
            assert np.all(F[:, -1] == 0.70710678118654757)
            if norm:
                assert np.allclose(np.sum(F[:, :-1] ** 2, axis=1), 0.5, rtol=1e-12)      # unit rows scaled by 1/sqrt 2
        else:
            assert np.all(F[:, -1] == 0.0) and np.array_equal(F[:, :-1], hk)              # raw hook rows, constant 0
        for mode in (0, 1, 2):
            rs, ok = o.dpp_relevance(rel, mode)
            assert ok
            L = o.dpp_kernel_matrix_f(F, rs, 0.7)
            r = [math.exp(0.7 * float(x)) for x in rs]
            for i in (0, 3, n - 1):
                for j in (0, 5, n - 1):
                    s_ = 0.0
                    for k in range(w + 1):
                        s_ = fma(float(F[i, k]), float(F[j, k]), s_)
                    assert L[i, j] == (r[i] * s_) * r[j]
            got = o.dpp_with_window(L, 10, 4).tolist()
            assert got == _np_dpp(L, 10, 4)
    # the table variant equals the older entry point that takes normalised embeddings
    e64 = o.l2_normalize_f64(emb.astype(np.float64))
    assert np.array_equal(o.dpp_kernel_matrix(e64, rel, 1.0), o.dpp_kernel_matrix_f(o.dpp_features(emb, None, True, True), rel, 1.0))
    # all-equal scores: both normalisations bail out as the reference does
    assert not o.dpp_relevance(np.full(5, 0.3), 1)[1] and not o.dpp_relevance(np.full(5, 0.3), 2)[1]


def test_go_float_format():
    assert [o.go_fmt_float(x) for x in (0.5, 1e21, 1.5e-7, 123456.0, 0.000123, 1e20, 100.0)] == \
        ["0.5", "1e+21", "1.5e-07", "123456", "0.000123", "1e+20", "100"]


def _py_ssd(emb, rel, gamma, topn, window, star):
    """Independent restatement of SSDWithSlidingWindow (ssd_sort.go:346-486) — plain Python floats,
    explicit queues as in the reference (utils.CycleQueue), fma chains for dot / norm."""
    from collections import deque
    E = [list(map(float, r)) for r in emb]
    N, d = len(E), len(E[0])
    if window <= 1:
        window = 5
    T = min(N, topn)

    def fma(x, y, s):            # correctly rounded x*y+s through exact rationals (no math.fma before 3.13)
        return float(Fraction(x) * Fraction(y) + Fraction(s))

    def dot(a, b):
        s = 0.0
        for x, y in zip(a, b):
            s = fma(x, y, s)
        return s

    def norm(a):
        return math.sqrt(dot(a, a))

    def max_idx(v):
        best, ind = float("nan"), 0
        for i, x in enumerate(v):
            if x != x:
                continue
            if x > best or best != best:
                best, ind = x, i
        return ind

    t = 1
    idx = max_idx(rel)
    selected = {idx}
    indices = [idx]
    volume = gamma
    if not star:
        l2 = norm(E[idx])
        if not (math.isnan(l2) or math.isinf(l2)):
            volume *= l2
    B, P = deque(), deque()
    while t < T:
        if t > window:
            i = B.popleft()
            proj = P.popleft()
            for j in range(N):
                if j in selected:
                    continue
                E[j] = [e + proj[j] * f for e, f in zip(E[j], E[i])]
        B.append(idx)
        proj = [0.0] * N
        den = dot(E[idx], E[idx])
        for j in range(N):
            if j in selected:
                continue
            p = dot(E[j], E[idx]) / den if den != 0 else float("nan")
            if math.isnan(p) or math.isinf(p):
                p = 1.0
            proj[j] = p
            E[j] = [e - p * f for e, f in zip(E[j], E[idx])]
        P.append(proj)
        t += 1
        q = []
        for i in range(N):
            if i in selected:
                q.append(-sys.float_info.max)
            else:
                l2 = norm(E[i])
                q.append(rel[i] + volume * (0.5 if (math.isnan(l2) or math.isinf(l2)) else l2))
        idx = max_idx(q)
        selected.add(idx)
        indices.append(idx)
        if not star:
            l2 = norm(E[idx])
            if not (math.isnan(l2) or math.isinf(l2)):
                volume *= l2
    return indices


def test_ssd_matches_python_restatement():
    _ssd_cases(True)


def _ssd_cases(exact):
    rng = np.random.default_rng(11)
    n, d = 60, 16
    emb = o.ssd_embeddings(rng.standard_normal((n, d)).astype(np.float32), True, True)
    rel = np.sort(rng.random(n))[::-1].copy()
    for topn, window, gamma, star in [(1, 5, 0.25, False), (7, 3, 0.25, False), (40, 5, 0.5, False),
                                      (40, 10, 0.25, True), (60, 4, 1.0, False), (200, 1, 0.25, False)]:
        got = o.ssd_window(emb, rel, gamma, topn, window, star).tolist()
        assert len(got) == min(n, topn) and len(set(got)) == len(got)
        if exact:
            assert got == _py_ssd(emb, rel.tolist(), gamma, topn, window, star), (topn, window, gamma, star)


def test_ssd_structure_and_numpy_cross_check():
    """Structure: the first pick is the best score, gamma → 0 degenerates to score order, and the
    second pick agrees with a direct numpy computation of one projection step."""
    rng = np.random.default_rng(12)
    n, d = 50, 16
    emb = o.ssd_embeddings(rng.standard_normal((n, d)).astype(np.float32), True, False)
    rel = np.sort(rng.random(n))[::-1].copy()
    assert o.ssd_window(emb, rel, 1e-12, 20, 5).tolist() == list(range(20))
    got = o.ssd_window(emb, rel, 0.5, 3, 5).tolist()
    assert got[0] == 0
    # second pick by hand: residuals after projecting out e_0, quality = rel + gamma*|e_0|*|residual|
    e0 = emb[0]
    res = emb - np.outer(emb @ e0 / (e0 @ e0), e0)
    q = rel + 0.5 * np.linalg.norm(e0) * np.linalg.norm(res, axis=1)
    q[0] = -np.inf
    assert got[1] == int(np.argmax(q))
    assert o.ssd_window(emb, rel, 0.5, 30, 5).tolist() != list(range(30))     # diversity changed the order


def test_ssd_quality_score_normalisation():
    rel = np.array([0.9, 0.7, 0.4, 0.1])
    z, ok = o.ssd_quality(rel, 1)
    assert ok and np.allclose(z, (rel - rel.mean()) / rel.std(), rtol=1e-14)
    m, ok = o.ssd_quality(rel, 2)
    assert ok and m[0] == 1.0 and abs(m[-1] - 1e-6) < 1e-18 and np.all(np.diff(m) < 0)
    assert o.ssd_quality(np.zeros(5), 1)[1] is False and o.ssd_quality(np.full(5, 0.3), 2)[1] is False
    assert o.ssd_quality(np.full(5, 0.3), 1)[1] is False                    # variance == 0
    assert np.array_equal(o.ssd_quality(rel, 0)[0], rel)


def test_features_map_reference_known_answers(golden):
    """web/features_map_test.go, transcribed: the request's `features` typing as the /api/recommend harness
    restates it (tools/http_harness.py:features_map)."""
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import http_harness as hh
    for c in golden["features_map"]:
        value, typ = hh.features_map(c["json"])[c["key"]]
        assert typ == c["type"], c["ref"]
        if "value" in c:
            assert value == c["value"], c["ref"]
    # parseNumber: beyond int64 falls to float64; mixed arrays stay untouched
    v, t = hh.features_map('{"big": 92233720368547758070, "mix": [1, [2]], "m": {"a": [1], "b": 2}}')["big"]
    assert t == "float64" and v == 9.223372036854776e19
    assert hh.features_map('{"mix": [1, true]}')["mix"][1] == "[]interface {}"
    assert hh.features_map('{"m": {"a": [1], "b": 2}}')["m"] == ({"a": ["1"], "b": ["2"]}, "map[string][]string")


def test_dnn3_blocked_form_equals_the_rowwise_form():
    """The oracle's rank forward runs 4 items per weight pass (AVX2 fma; the CPU baseline's rank leg); every output is
    still its own k-ascending fmaf chain, so the bits are those of the one-item-at-a-time form (threads < 0 selects it)."""
    w = o.Dnn3Weights()
    rng = np.random.default_rng(12)
    items = rng.standard_normal((1003, 128)).astype(np.float32)
    u = rng.standard_normal(128).astype(np.float32)
    for prec in (0, 1):
        a = o.dnn3_forward(w, prec, u, items, threads=4)
        b = o.dnn3_forward(w, prec, u, items, threads=-4)
        assert np.array_equal(np.asarray(a).view(np.uint64 if a.dtype == np.float64 else np.uint32),
                              np.asarray(b).view(np.uint64 if b.dtype == np.float64 else np.uint32))


def test_expr_pow_follows_gos_integer_power_loop():
    """math.Pow (math/pow.go) applies the integer part of an exponent by repeated squaring of Frexp(x)'s mantissa with the
    binary exponent beside it: exact where the products are, and its own value where they are not — the restatement must not
    fall back on a libm pow for integer-valued exponents."""
    assert o.go_pow(400.0, 4.0) == 25600000000.0
    assert o.go_pow(10.0, 308.0) == 1.0000000000000006e308          # (the loop's result; a correctly rounded pow gives 1e308)
    assert o.go_pow(10.0, 309.0) == float("inf") and o.go_pow(10.0, -330.0) == 0.0
    assert o.go_pow(-2.0, 3.0) == -8.0 and o.go_pow(-2.0, 4.0) == 16.0
    assert o.go_pow(-238.9, 25599992000.0) == float("inf") and o.go_pow(-238.9, 25599992001.0) == float("-inf")
    assert o.go_pow(7.25, 1.0) == 7.25 and o.go_pow(1.5, -3.0) == 1.0 / (1.5 * 1.5 * 1.5)
    assert o.expr_eval(o.expr_parse("4e2^4e0%1000"), lambda name: None) == 0.0


def test_pow_last_ulp_classifier():
    """oracle.pow_last_ulp_explains (the soaks' classifier): a power inside the exponent of a negative base — the inner pow's last ulp
    decides whether the outer exponent is an integer, i.e. a number or NaN — is explained; an arbitrary wrong value is not."""
    src = "(${ctr})*(2.50*0 - 3e1)^(3.864*${ctr} ^ 9.650 * 886^${current_score}#${cvr} # ${a_b}+396)"
    vals = {"ctr": 0.49297272577611595, "current_score": 5.956023815646717, "cvr": 23.186559967770872, "a_b": 1280679.6164711325}
    ast = o.expr_parse(src)
    ev = lambda: o.expr_eval(ast, lambda nm: vals.get(nm))
    assert math.isnan(ev())
    assert o.pow_last_ulp_explains(ev, float("inf")) and not o.pow_last_ulp_explains(ev, 1.0)
    src = "maxIndex(${probs})*0 + (maxValue(${probs}))^(213+592^4.723*141) / (-9.876 * 369)-474"
    data = {"probs": [-0.29112667712280665, -1.0701016109674397, -0.8296780524853117]}
    tree = o.antlr_parse(src)
    ev = lambda: o.antlr_result(tree, data)
    assert ev() == -474.0 and o.pow_last_ulp_explains(ev, float("nan")) and not o.pow_last_ulp_explains(ev, 5.0)
    assert o.go_pow(2.0, 10.0) == 1024.0                                             # (the hook is restored)


def test_mixture_rows_are_clustered_normalised_and_reproducible():
    """o.synth_mixture_rows (oracle.c: orc_synth_mixture_rows; the device's pg_table_fill_mixture regenerates it bit for bit — GPU
    test tests/test_gpu_i4m.py): rows of norm 1, any slice equals the same rows of a larger draw, a row's nearest centre is its own,
    and queries (stream 1) are new points of the SAME centres."""
    seed, C_, sigma, d = 0x5EED0007, 50, 0.1, 128
    a = o.synth_mixture_rows(seed, 0, 6000, d, C_, sigma)
    b = o.synth_mixture_rows(seed, 1234, 100, d, C_, sigma)
    assert np.array_equal(a[1234:1334].view(np.uint32), b.view(np.uint32))
    assert np.max(np.abs(np.linalg.norm(a.astype(np.float64), axis=1) - 1.0)) < 1e-6     # (fp32 normalisation)
    cen = o.synth_rows(seed + 1, 0, C_, d)                       # the centres: SURVEY 8d's normalised rows of seed + 1
    sim = a.astype(np.float64) @ cen.astype(np.float64).T
    own = sim.max(axis=1)
    assert np.all(own > 0.98) and np.all(np.sort(sim, axis=1)[:, -2] < 0.6)      # cos to the own centre ~ 1 / sqrt(1 + sigma^2)
    q = o.synth_mixture_rows(seed, 0, 200, d, C_, sigma, stream=1)
    assert not np.array_equal(q[:200], a[:200])
    assert np.all((q.astype(np.float64) @ cen.astype(np.float64).T).max(axis=1) > 0.98)
    counts = np.bincount(sim.argmax(axis=1), minlength=C_)
    assert counts.min() > 60 and counts.max() < 200                               # 6 000 rows over 50 centres
