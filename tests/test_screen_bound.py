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


# ---- round 6: the 4-bit matrix-pipe screen (csrc/recall_i4m.hip) and the two-digit refinement (csrc/recall_r2.hip), restated --------
def bf16_up(x):
    """smallest bf16 value >= x (x >= 0), as float32 (recall_i4.hip: bf16_up)"""
    b = np.float32(x).view(np.uint32)
    b = np.where(b & np.uint32(0xFFFF), (b | np.uint32(0xFFFF)) + np.uint32(1), b).astype(np.uint32)
    return b.view(np.float32)


def quantize_table4(tab):
    """table_quant4_kernel: X in [-7, 7] with ONE scale per row (bf16, rounded up, never below 1e-30), R_r >= ||x - s_r X|| (bf16, up)"""
    amax = np.abs(tab).max(axis=1)
    s = bf16_up(np.maximum(amax / np.float32(7.0), np.float32(1e-30))).astype(np.float64)
    X = np.clip(np.rint(tab.astype(np.float64) / s[:, None]), -7, 7)
    resid = np.sqrt(((tab.astype(np.float64) - s[:, None] * X) ** 2).sum(axis=1))
    R = bf16_up((resid * 1.001 + 1e-30).astype(np.float32)).astype(np.float64)
    return s, X, R


@pytest.mark.parametrize("case", CASES + ["zero_rows", "one_hot_rows"])
def test_4bit_row_bound_and_its_integer_form_hold(case):
    """|<x, q> - s_r s_q <X, Q>| <= R_r ||q|| + H_r ||q - q^||, H_r = min(7 sqrt(d) s_r, N + R4); and the form screen4m_kernel
    evaluates — I >= tau u_r - (rho_r + eta_r kappa) beta_q - 8 with the batch-wide kappa >= alpha_q / beta_q — never drops a
    (row, query) pair whose exact score reaches the threshold, for thresholds of either sign."""
    rng = np.random.default_rng((hash(case) + 6) % 2**32)
    n, d = 3000, 128
    tab = rng.standard_normal((n, d)) * 0.3
    if case == "uniform":
        tab = rng.uniform(-1, 1, (n, d))
    elif case == "outlier":
        tab[:, 3] *= 50.0
    elif case == "tiny_rows":
        tab[: n // 2] *= 1e-6
        tab[:10] *= 1e-30
    elif case == "zero_rows":
        tab[::3] = 0.0
    elif case == "one_hot_rows":
        tab[:] = 0.0
        tab[np.arange(n), rng.integers(0, d, n)] = rng.standard_normal(n)
    tab = tab.astype(np.float32)
    s, X, R = quantize_table4(tab)
    norm = np.sqrt((tab.astype(np.float64) ** 2).sum(axis=1)).max() * 1.0001
    h_cap = (norm + R.max() * 1.000001 + 1e-6 * norm) * 1.000001
    H = np.minimum(79.1961 * s, h_cap)
    qs = rng.standard_normal((16, d)).astype(np.float32)
    qs[0] = 0.0
    qs[1] = 0.0
    qs[1, 3] = 1.0
    if case == "aligned":
        for j in range(2, 10):
            r = tab[j].astype(np.float64) - s[j] * X[j]
            qs[j] = (r / max(np.abs(r).max(), 1e-30)).astype(np.float32)
    consts = []
    for q in qs:
        s_q, Q, dq = quantize_query(q)
        nq = np.sqrt((q.astype(np.float64) ** 2).sum())
        B = nq * (1.0 + 1e-5) * 1.000001 + 1e-30
        A = dq * 1.0001 + 1e-5 * nq + 1e-30
        consts.append((s_q, Q, B, A))
        exact = tab.astype(np.float64) @ q.astype(np.float64)
        approx = s * s_q * (X @ Q)
        assert np.all(np.abs(exact - approx) <= R * B + H * A), case
    kappa = max(A / B for _, _, B, A in consts) * (1 + 1e-6)
    u = 1.0 / s
    m = (R * u + np.minimum(79.1962, h_cap * u) * kappa) * 1.000001
    for q, (s_q, Q, B, A) in zip(qs, consts):
        exact = tab.astype(np.float64) @ q.astype(np.float64)
        beta = B / s_q * (1 + 2e-6)
        for thr in (np.quantile(exact, 0.99), exact.max(), 0.0, np.quantile(exact, 0.3)):
            T = (thr / s_q) * u - m * beta - 8.0
            keep = (X @ Q) >= T
            assert np.all(keep[exact >= thr]), (case, thr)


@pytest.mark.parametrize("case", CASES)
def test_two_digit_refinement_bound_holds_and_is_two_hundred_times_tighter(case):
    """x = s8 X8 + s8r Xr + e2 (s8r = s8 / 254), q = sq16 Q16 + eq (|Q16| <= 32639):
    |<x, q> - sq16 (s8 <X8, Q16> + s8r <Xr, Q16>)| <= (N + R2) ||eq|| + R2 ||q||, and Q16 = 256 Qh + Ql with both halves in int8."""
    rng = np.random.default_rng((hash(case) + 16) % 2**32)
    n, d = 3000, 128
    tab = rng.standard_normal((n, d)) * 0.3
    if case == "uniform":
        tab = rng.uniform(-1, 1, (n, d))
    elif case == "outlier":
        tab *= 0.2
        tab[17, 3] = 40.0
    elif case == "tiny_rows":
        tab[: n // 2] *= 1e-6
    tab = tab.astype(np.float32)
    s8, X8, resid, norm = quantize_table(tab)
    s8 = float(np.float32(s8))
    s8r = float(np.float32(s8) / np.float32(254.0))
    r = tab.astype(np.float64) - s8 * X8
    Xr = np.clip(np.rint(r / s8r), -127, 127)
    e2 = r - s8r * Xr
    R2 = np.sqrt((e2 ** 2).sum(axis=1)).max() * 1.01 + 1e-7 * norm
    assert R2 < resid / 100 or case == "tiny_rows"
    qs = rng.standard_normal((12, d)).astype(np.float32)
    qs[0] = 0.0
    if case == "aligned":
        for j in range(2, 8):
            qs[j] = (e2[j] / max(np.abs(e2[j]).max(), 1e-300)).astype(np.float32)
    for q in qs:
        sq = max(float(np.abs(q).max()) / 32639.0, 1e-30)
        Q16 = np.clip(np.rint(q.astype(np.float64) / sq), -32639, 32639)
        Qh = np.floor((Q16 + 128) / 256)
        Ql = Q16 - 256 * Qh
        assert Qh.min() >= -128 and Qh.max() <= 127 and Ql.min() >= -128 and Ql.max() <= 127
        eq = np.sqrt(((q.astype(np.float64) - sq * Q16) ** 2).sum())
        nq = np.sqrt((q.astype(np.float64) ** 2).sum())
        eps2 = ((norm + R2) * eq + R2 * nq) * 1.0001 + 1e-5 * norm * nq + 1e-30
        exact = tab.astype(np.float64) @ q.astype(np.float64)
        M = sq * (s8 * (X8 @ Q16) + s8r * (Xr @ Q16))
        assert np.abs(exact - M).max() <= eps2, (case, np.abs(exact - M).max(), eps2)
        if nq > 0:
            s_q, Q, dq = quantize_query(q)
            # (the spec-chain slack 1e-5 N ||q|| is common to both: the data-dependent part is ~1/250 of the int8 screen's)
            assert eps2 - 1e-5 * norm * nq < (eps_of(q, resid, norm, dq) - 1e-5 * norm * nq) / 100 or case == "tiny_rows"
