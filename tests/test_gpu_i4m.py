"""The mid-batch recall (csrc/recall_i4m.hip): passes of 1 … 64 queries stream the 4-bit shadow through the int8 matrix
pipe, thin the suspects on the int8 shadow, re-score exactly.  Results must stay bit-identical to the oracle's
(`o.recall_topk`: the reference's VectorRecall → FaissModel.Run top-K, service/recall/vector_recall.go:32-123) whatever
the screens let through, on data built against the bounds, and the pass must really have read the narrow shadow."""
import numpy as np
import pytest

import pairec_amd as pa
from oracle import oracle as o

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


@pytest.fixture
def small_table_opts(ctx):
    """the 4-bit screens on tables far below their production size limit, a quarter of the rows as the pilot sample"""
    for name, v in (("i4_min_rows", "0"), ("pilot_fraction", "0.25"), ("i4m_max_lambda", "1000"), ("i4_max_lambda", "3"),
                    ("i4m_max_pairs", "1e12")):
        ctx.set_option(name, v)
    yield ctx
    for name, v in (("i4_min_rows", str(1 << 22)), ("pilot_fraction", "0"), ("i4m_max_lambda", "2.2"), ("i4_max_lambda", "1.7"),
                    ("no_screen_i4m", "0"), ("i4m_max_queries", "64"), ("i4m_max_pairs", "2.4e7")):
        ctx.set_option(name, v)


def check(ctx, t, tab, q, k, expect_narrow=True):
    rows, scores, cnt = t.recall_topk(q, k)
    _, nbytes = ctx.last_scan_kernel()
    orow, osc = o.recall_topk(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    assert cnt.tolist() == [min(k, tab.shape[0])] * q.shape[0]
    if expect_narrow is not None and bool(np.all(np.isfinite(q))):
        # less than one pass over the int8 shadow (128 B per row) <=> the full pass streamed 68 B per row
        assert (nbytes < tab.shape[0] * 128) == expect_narrow, (nbytes, tab.shape[0] * 128, expect_narrow)


@pytest.mark.parametrize("nq", [1, 2, 3, 5, 8, 16, 31, 32, 33, 48, 64])
def test_mid_batch_uniform_rows(small_table_opts, nq):
    """the benchmark's row distribution (uniform, normalised), every batch size class incl. both ends of each query block"""
    ctx = small_table_opts
    n, d, k = 333_333, 128, 500                  # ragged against the 64-row pieces
    t = pa.Table(ctx, n, d, row_offset=3)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 3, n, d)
    q = o.synth_rows(o.SEED_QUERY, 11, nq, d)
    rows, scores, cnt = t.recall_topk(q, k)
    _, nbytes = ctx.last_scan_kernel()
    orow, osc = o.recall_topk(tab, q, k, row_offset=3)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    assert cnt.tolist() == [k] * nq
    assert nbytes < n * 128
    t.destroy()


def test_mid_batch_hostile_data(small_table_opts):
    """an outlier inside a row (coarsens that row's 4-bit scale), tiny rows, zero rows, winners that differ by far less
    than a 4-bit step, exact duplicates, a zero query, one-hot, tiny, huge and non-finite queries"""
    ctx = small_table_opts
    rng = np.random.default_rng(291)
    n, d, k = 400_000, 128, 300
    tab = rng.standard_normal((n, d)).astype(np.float32) * 0.05
    tab[1234, 17] = 0.5
    tab[2000:2600] *= 1e-6
    tab[2600:2700] *= 1e-33                      # below the shadow's scale floor
    tab[3000:3100] = 0.0
    near = (rng.standard_normal(d) * 0.05).astype(np.float32)
    tab[5000:5400] = near * (1.0 + 1e-6 * np.arange(400, dtype=np.float32)[:, None])
    tab[7000:7050] = tab[6000:6050]
    qs = rng.standard_normal((64, d)).astype(np.float32)
    qs[0] = 0.0
    qs[1] = 0.0
    qs[1, 17] = 1.0
    qs[2] = near
    qs[3] = -near
    qs[6] *= np.float32(1e-20)
    qs[7] *= np.float32(1e15)
    qs[8] *= np.float32(1e-36)
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    assert t.screen_info()[0] == 1
    for lo, hi in ((0, 5), (0, 9), (2, 40), (0, 64), (9, 14), (20, 53), (2, 3), (3, 4), (6, 8), (7, 8)):
        # (a zero query makes every row a suspect: its plan overflows by design and the next plan answers on the wide shadow)
        check(ctx, t, tab, qs[lo:hi], k, expect_narrow=None if lo < 2 else True)
    bad = qs[10:20].copy()
    bad[3, 5] = np.inf
    bad[7, 9] = np.nan
    check(ctx, t, tab, bad, k)
    # the shadow follows uploads and swaps
    tab2 = tab.copy()
    tab2[100_000:160_000] = rng.standard_normal((60_000, d)).astype(np.float32) * 0.2
    t.upload(tab2[100_000:160_000], row0=100_000)
    check(ctx, t, tab2, qs[9:30], k)
    other = pa.Table(ctx, n, d)
    tab3 = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    other.upload(tab3)
    check(ctx, other, tab3, qs[9:20], k)
    t.swap(other)
    check(ctx, t, tab3, qs[9:20], k)
    check(ctx, other, tab2, qs[30:64], k)
    # switched off, and above its batch limit: the int8 shadow serves
    ctx.set_option("no_screen_i4m", "1")
    check(ctx, t, tab3, qs[9:20], k, expect_narrow=False)
    ctx.set_option("no_screen_i4m", "0")
    ctx.set_option("i4m_max_queries", "16")
    check(ctx, t, tab3, qs[9:26], k, expect_narrow=False)
    check(ctx, t, tab3, qs[9:25], k, expect_narrow=True)
    ctx.set_option("i4m_max_queries", "64")
    t.destroy()
    other.destroy()


def test_mid_batch_heavy_tails_ties_and_crowds(small_table_opts):
    """heavy-tailed elements (int8-range Student-t: every term of the 4-bit bound is relative to the row), a table of heavy
    ties, and rows crowded within the screens' error of the K-th score (the suspect lists overflow: the fallback answers)"""
    ctx = small_table_opts
    rng = np.random.default_rng(292)
    n, d, k = 300_011, 128, 200
    heavy = (rng.standard_t(2.2, (n, d)) * 0.01).astype(np.float32)
    np.clip(heavy, -0.4, 0.4, out=heavy)
    heavy[-5:] *= 2.0                              # winners in the ragged last rows
    t = pa.Table(ctx, n, d)
    t.upload(heavy)
    q = rng.standard_normal((24, d)).astype(np.float32)
    if t.screen_info()[0] == 1:
        check(ctx, t, heavy, q, k)
    else:
        check(ctx, t, heavy, q, k, expect_narrow=None)
    ties = np.zeros((n, d), dtype=np.float32)
    ties[:, 0] = (np.arange(n) % 7).astype(np.float32)
    ties[:, 1:] = (rng.standard_normal((1, d - 1)) * 0.01).astype(np.float32)
    t.upload(ties)
    check(ctx, t, ties, q[:12], k, expect_narrow=None)
    base = rng.standard_normal(d).astype(np.float32)
    crowd = (base[None, :] * (1.0 + 1e-5 * rng.standard_normal((n, 1)))).astype(np.float32)
    crowd += (1e-4 * rng.standard_normal((n, d))).astype(np.float32)
    t.upload(crowd)
    qc = (base[None, :] + 0.01 * rng.standard_normal((10, d))).astype(np.float32)
    check(ctx, t, crowd, qc, k, expect_narrow=None)
    t.destroy()


def test_mid_batch_where_clause_and_l2(small_table_opts):
    """a WhereClause evaluated in place rides the same pass (the predicate is applied where candidates are made); the
    squared-Euclidean order stays on the int8 shadow — both exact (hologres_vector_recall.go:49-62, _v2.go:23)"""
    ctx = small_table_opts
    rng = np.random.default_rng(293)
    n, d, k, nq = 280_000, 128, 150, 20
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    q = o.synth_rows(o.SEED_QUERY, 5, nq, d)
    col = rng.integers(0, 10, n).astype(np.int32)
    fs = pa.Features(ctx, n)
    fs.set_column("cat", pa.F_I32, col)
    rows, scores, cnt = t.recall_topk_where(fs, "cat", ">=", 3, q, k)
    _, nbytes = ctx.last_scan_kernel()
    keep = np.nonzero(col >= 3)[0]
    orow, osc = o.recall_topk(tab[keep], q, k)
    assert np.array_equal(rows, keep[orow.astype(np.int64)].astype(np.uint64)) and np.array_equal(bits(scores), bits(osc))
    assert nbytes < n * 128
    rows, dist, _ = t.recall_topk_l2(q, k)
    orow, od = o.recall_topk_l2(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(dist), bits(od))
    fs.destroy()
    t.destroy()


def test_mid_batch_matches_single_requests(small_table_opts):
    """a request's answer does not depend on the pass it rode in: 1 (4-bit VALU screen), 12 and 40 (4-bit matrix-pipe
    screen) and 200 queries (int8 screen) per pass give the same bits"""
    ctx = small_table_opts
    n, d, k = 260_000, 128, 400
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    q = o.synth_rows(o.SEED_QUERY, 0, 200, d)
    rows, scores, _ = t.recall_topk(q, k)
    for lo, hi in ((0, 12), (12, 52), (150, 200)):
        r, s, _ = t.recall_topk(q[lo:hi], k)
        assert np.array_equal(r, rows[lo:hi]) and np.array_equal(bits(s), bits(scores[lo:hi]))
    for i in (0, 77):
        r, s, _ = t.recall_topk(q[i:i + 1], k)
        assert np.array_equal(r[0], rows[i]) and np.array_equal(bits(s[0]), bits(scores[i]))
    t.destroy()


@pytest.mark.parametrize("sigma,nq", [(0.3, 256), (0.1, 256), (0.03, 130), (0.1, 24), (0.03, 3)])
def test_clustered_rows_are_exact(ctx, sigma, nq):
    """pg_table_fill_mixture: 20 centres on the sphere, 2 M rows = 100 K per cluster, queries drawn from the same mixture — a
    query's top-K sits inside one cluster whose members score within the screens' error of each other (9-20 suspects per
    answer at 256 queries; the hit-record areas grow with the table instead of the pass falling to the exact scan).  The fill is
    the oracle's bit for bit, and ids / order / score bits of every batch size class equal the oracle's."""
    n, d, k, C_, seed = 2_000_000, 128, 5000, 20, 0x5EED0007
    t = pa.Table(ctx, n, d)
    t.fill_mixture(seed, C_, sigma)
    tab = o.synth_mixture_rows(seed, 0, n, d, C_, sigma)
    assert np.array_equal(bits(t.download(n - 70_000, 70_000)), bits(tab[n - 70_000:]))
    q = o.synth_mixture_rows(seed, 555, nq, d, C_, sigma, stream=1)
    ctx.set_option("i4_min_rows", "0")
    try:
        for rep in range(2):                               # (the second batch runs with the areas the first one grew)
            s0 = ctx.stats()
            rows, scores, cnt = t.recall_topk(q, k)
            s1 = ctx.stats()
            if rep == 0:
                orow, osc = o.recall_topk(tab, q, k)
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc)) and cnt.tolist() == [k] * nq
        assert s1.recall_screen_overflows == s0.recall_screen_overflows        # the steady state stays on the screened pass
        assert ctx.last_scan_kernel()[1] <= n * 128 * 1.3
        if nq > 64:
            # crowded rows switch the refinement stage on (csrc/recall_r2.hip): the third batch re-scores a fraction of the
            # int8 screen's suspects — and still answers with the oracle's bits
            s0 = ctx.stats()
            rows, scores, cnt = t.recall_topk(q, k)
            s1 = ctx.stats()
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
            susp, resc = s1.recall_suspects - s0.recall_suspects, s1.recall_rescored - s0.recall_rescored
            assert susp > 3 * k * nq and resc < susp / 2, (susp, resc)
            ctx.set_option("no_r2", 1)
            s0 = ctx.stats()
            rows, scores, cnt = t.recall_topk(q, k)
            s1 = ctx.stats()
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
            assert s1.recall_rescored - s0.recall_rescored == s1.recall_suspects - s0.recall_suspects
    finally:
        ctx.set_option("i4_min_rows", str(1 << 22))
        ctx.set_option("no_r2", 0)
    t.destroy()


def test_refinement_stage_on_hostile_data(ctx):
    """The two-digit refinement (int8 + int8 residual shadow, 16-bit queries: csrc/recall_r2.hip) forced on for every
    int8-screened pass (r2_min_factor 0) over the data built against the int8 bound: an outlier that coarsens the table's
    scale, tiny rows, zero rows, winners that differ by less than a quantisation step, exact duplicates, zero / one-hot /
    tiny / huge / non-finite queries, negative best scores — ids, order and score bits stay the oracle's."""
    rng = np.random.default_rng(77)
    n, d, k, nq = 150_000, 128, 600, 100
    tab = rng.standard_normal((n, d)).astype(np.float32) * 0.05
    tab[1234, 17] = 0.5
    tab[2000:2600] *= 1e-6
    tab[3000:3100] = 0.0
    near = (rng.standard_normal(d) * 0.05).astype(np.float32)
    tab[5000:5400] = near * (1.0 + 1e-6 * np.arange(400, dtype=np.float32)[:, None])
    tab[7000:7050] = tab[6000:6050]
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = 0.0
    q[1] = 0.0
    q[1, 17] = 1.0
    q[2] = near
    q[3] = -near
    q[4] *= np.float32(1e-20)
    q[5] *= np.float32(1e15)
    q[6:30:3] *= -1.0
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    assert t.screen_info()[0] == 1
    ctx.set_option("r2_min_factor", -1)
    try:
        for lo, hi in ((0, 100), (7, 80), (0, 3)):
            s0 = ctx.stats()
            rows, scores, _ = t.recall_topk(q[lo:hi], k)
            s1 = ctx.stats()
            orow, osc = o.recall_topk(tab, q[lo:hi], k)
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc)), (lo, hi)
            if lo == 7:
                assert 0 < s1.recall_rescored - s0.recall_rescored <= s1.recall_suspects - s0.recall_suspects
        bad = q[10:90].copy()
        bad[3, 5] = np.inf
        bad[7, 9] = np.nan
        rows, scores, _ = t.recall_topk(bad, k)
        orow, osc = o.recall_topk(tab, bad, k)
        assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
        # the residual shadow follows an upload
        tab[40_000:90_000] = rng.standard_normal((50_000, d)).astype(np.float32) * 0.2
        t.upload(tab[40_000:90_000], row0=40_000)
        rows, scores, _ = t.recall_topk(q[7:99], k)
        orow, osc = o.recall_topk(tab, q[7:99], k)
        assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    finally:
        ctx.set_option("r2_min_factor", 3)
    t.destroy()
