"""The int8 screen's error bound, restated in numpy (no GPU): for every row x and query q

    |<x, q> - s_x s_q <X, Q>|  <=  R ||q|| + (N + R) ||q - q^||

with X = clamp(rint(x / s_x), +-127), s_x = max|table| / 127, Q likewise per query, R = max row residual
||x - s_x X||, N = max row norm — the inequality recall.hip's screen_prep8_kernel / screen_thr8_kernel rely on
(DESIGN.md 4.1a).  Checked on random data and on data built to approach the bound (residuals aligned with the
query), and the integer cutoff is checked never to exclude a row whose exact score reaches the threshold."""
import numpy as np
import pytest


def quantize_table(tab):
    s = max(float(np.abs(tab).max()) / 127.0, 1e-30)
    X = np.clip(np.rint(tab.astype(np.float64) / s), -127, 127)
    resid = np.sqrt(((tab.astype(np.float64) - s * X) ** 2).sum(axis=1)).max()
    norm = np.sqrt((tab.astype(np.float64) ** 2).sum(axis=1)).max()
    return s, X, resid, norm


def quantize_query(q):
    s = max(float(np.abs(q).max()) / 127.0, 1e-30)
    Q = np.clip(np.rint(q.astype(np.float64) / s), -127, 127)
    dq = np.sqrt(((q.astype(np.float64) - s * Q) ** 2).sum())
    return s, Q, dq


def eps_of(q, resid, norm, dq):
    nq = np.sqrt((q.astype(np.float64) ** 2).sum())
    return (resid * nq + (norm + resid) * dq) * 1.0001 + 1e-5 * norm * nq + 1e-30


CASES = ["uniform", "gauss", "outlier", "tiny_rows", "aligned"]


@pytest.mark.parametrize("case", CASES)
def test_int8_bound_holds(case):
    rng = np.random.default_rng(hash(case) % 2**32)
    n, d = 4000, 128
    if case == "uniform":
        tab = rng.uniform(-1, 1, (n, d))
    elif case == "gauss":
        tab = rng.standard_normal((n, d)) * 0.3
    elif case == "outlier":
        tab = rng.standard_normal((n, d)) * 0.05
        tab[17, 3] = 4.0e4
    elif case == "tiny_rows":
        tab = rng.standard_normal((n, d))
        tab[: n // 2] *= 1e-6
    else:
        tab = rng.uniform(-1, 1, (n, d))
    tab = tab.astype(np.float32)
    s_x, X, resid, norm = quantize_table(tab)
    qs = rng.standard_normal((24, d)).astype(np.float32)
    qs[0] = 0.0
    qs[1] = 0.0
    qs[1, 3] = 1.0
    if case == "aligned":
        # queries proportional to a row's own quantisation residual: Cauchy-Schwarz tight in the table-side term
        for j in range(2, 12):
            r = tab[j].astype(np.float64) - s_x * X[j]
            qs[j] = (r / max(np.abs(r).max(), 1e-30)).astype(np.float32)
    for q in qs:
        s_q, Q, dq = quantize_query(q)
        eps = eps_of(q, resid, norm, dq)
        exact = tab.astype(np.float64) @ q.astype(np.float64)
        approx = s_x * s_q * (X @ Q)
        err = np.abs(exact - approx)
        assert err.max() <= eps, (case, err.max(), eps)
        # the integer cutoff of screen_thr8_kernel never drops a row whose exact score reaches the threshold
        for thr in (np.quantile(exact, 0.99), exact.max(), 0.0):
            T = np.floor((thr - eps) / (s_x * s_q)) - 1.0
            keep = (X @ Q) >= T
            assert np.all(keep[exact >= thr]), (case, thr)


def test_bound_is_not_vacuous_on_benchmark_like_data():
    """On uniform rows (the benchmark's synthetic table) the margin is a small fraction of the score spread, so the
    screen actually screens: fewer than 2 % of rows pass a cutoff placed at the 99.9th percentile."""
    rng = np.random.default_rng(3)
    tab = rng.uniform(-1, 1, (20000, 128)).astype(np.float32)
    q = rng.uniform(-1, 1, 128).astype(np.float32)
    s_x, X, resid, norm = quantize_table(tab)
    s_q, Q, dq = quantize_query(q)
    eps = eps_of(q, resid, norm, dq)
    exact = tab.astype(np.float64) @ q.astype(np.float64)
    thr = np.quantile(exact, 0.999)
    T = np.floor((thr - eps) / (s_x * s_q)) - 1.0
    frac = np.mean((X @ Q) >= T)
    assert eps < 0.15 * exact.std()
    assert frac < 0.02
