"""GPU parity tests: every stage of the hot path, through the C ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact for row ids / ordering / fp32 recall scores / PG_PREC_F32
pre-activations; the stated tolerance where a transcendental (expf, exp, pow) or the bf16 MFMA's
internal accumulation order is involved."""
import math
import os

import numpy as np
import pytest

import pairec_amd as pa
from oracle import oracle as o

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


# ---------------------------------------------------------------------------------------------
# tables
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d", [(1, 64), (1000, 64), (70001, 128), (5000, 256)])
def test_table_fill_bitexact(ctx, n, d):
    t = pa.Table(ctx, n, d, row_offset=12345)
    t.fill_synthetic(o.SEED_TABLE)
    ref = o.synth_rows(o.SEED_TABLE, 12345, n, d)
    assert np.array_equal(bits(t.download(0, n)), bits(ref))
    t.fill_synthetic(o.SEED_TABLE, normalize=False)
    assert np.array_equal(bits(t.download(0, n)), bits(o.synth_rows(o.SEED_TABLE, 12345, n, d, normalize=False)))
    t.destroy()


def test_table_upload_gather_swap(ctx):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((300, 64)).astype(np.float32)
    b = rng.standard_normal((300, 64)).astype(np.float32)
    ta, tb = pa.Table(ctx, 300, 64), pa.Table(ctx, 300, 64)
    ta.upload(a)
    tb.upload(b)
    idx = [0, 299, 17, 17, 5]
    assert np.array_equal(ta.gather(idx), a[idx])            # VectorDao lookup analogue
    ta.swap(tb)                                              # hot swap (hologres partition switch)
    assert np.array_equal(ta.gather(idx), b[idx]) and np.array_equal(tb.download(0, 300), a)
    with pytest.raises(pa._lib.PgError):
        ta.gather([300])
    ta.destroy()
    tb.destroy()


# ---------------------------------------------------------------------------------------------
# recall
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,k,nq", [
    (1000, 64, 200, 1),          # cfg 1 shape in miniature (dot-product top-200)
    (40000, 128, 200, 3),        # two chunks, one block per wave
    (140000, 128, 500, 32),      # three chunks, several blocks per wave, full 32-query block
    (140000, 128, 300, 64),      # > 32 queries: int8 screen + exact re-scoring
    (70000, 64, 100, 45),
    (250000, 128, 5000, 128),    # full 128-query pass, K=5000
    (200000, 128, 2000, 256),    # 256 queries: two query halves per wave, two waves per SIMD
    (2_300_000, 128, 12000, 40), # pilot plan with K > 8192: select beyond its LDS list, LDS bitonic final sort
    (90000, 64, 300, 200),
    (100003, 128, 1000, 100),
    (300017, 64, 5000, 7),       # ragged row count, K=5000
    (123457, 192, 16384, 2),     # maximum K, dim 192
    (90000, 256, 50, 5),
])
def test_recall_matches_oracle_bitexact(ctx, n, d, k, nq):
    t = pa.Table(ctx, n, d, row_offset=7)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 7, n, d)
    q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    rows, scores, cnt = t.recall_topk(q, k)
    orow, osc = o.recall_topk(tab, q, k, row_offset=7)
    assert cnt.tolist() == [min(k, n)] * nq
    assert np.array_equal(rows, orow)                        # ids and order: exact
    assert np.array_equal(bits(scores), bits(osc))           # scores: bit-exact
    t.destroy()


def test_recall_batching_invariance(ctx):
    """A request's result must not depend on what it is batched with (k-ordered chain spec)."""
    n, d, k = 60000, 128, 300
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    q = o.synth_rows(o.SEED_QUERY, 0, 300, d)                # 300 → two table passes (256 + 44)
    rows, scores, _ = t.recall_topk(q, k)
    for i in (0, 13, 31, 32, 63, 64, 127, 128, 255, 256, 299):
        r1, s1, _ = t.recall_topk(q[i:i + 1], k)
        assert np.array_equal(r1[0], rows[i]) and np.array_equal(bits(s1[0]), bits(scores[i]))
    t.destroy()


def test_recall_edge_cases(ctx):
    d = 64
    # k > rows: tail padded with row=UINT64_MAX, score=-inf
    tab = o.synth_rows(o.SEED_TABLE, 0, 10, d)
    t = pa.Table(ctx, 10, d)
    t.upload(tab)
    q = o.synth_rows(o.SEED_QUERY, 0, 2, d)
    rows, scores, cnt = t.recall_topk(q, 16)
    orow, osc = o.recall_topk(tab, q, 16)
    assert cnt.tolist() == [10, 10]
    assert np.array_equal(rows[:, :10], orow) and np.array_equal(bits(scores[:, :10]), bits(osc))
    assert np.all(rows[:, 10:] == np.uint64(2 ** 64 - 1)) and np.all(np.isneginf(scores[:, 10:]))
    t.destroy()
    # heavy ties + signed zeros + a NaN row: order = score desc (totalOrder, NaN last), row asc
    n = 50000
    tab = np.zeros((n, d), dtype=np.float32)
    tab[:, 0] = (np.arange(n) % 7).astype(np.float32)
    tab[100, 0] = np.nan
    tab[200, 0] = -0.0
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    q = np.zeros((1, d), dtype=np.float32)
    q[0, 0] = 1.0
    for k in (10, 8000):
        rows, scores, _ = t.recall_topk(q, k)
        orow, osc = o.recall_topk(tab, q, k)
        assert np.array_equal(rows, orow)
        assert np.array_equal(bits(scores), bits(osc))
    t.destroy()


def test_recall_screen_is_exact_on_hostile_data(ctx):
    """The screen (int8 at this width) must never lose a true top-K member: unnormalised rows with a wide range of
    norms, clustered near-duplicates (scores that differ only below bf16 resolution), heavy ties and
    negative scores, scanned with 48 queries (screened path) — ids, order and score bits exact."""
    rng = np.random.default_rng(11)
    n, d, k, nq = 180_000, 128, 800, 48
    base = rng.standard_normal((64, d)).astype(np.float32)
    tab = base[rng.integers(0, 64, n)] * (1 + 1e-4 * rng.standard_normal((n, 1))).astype(np.float32)
    tab *= (10.0 ** rng.uniform(-2, 2, (n, 1))).astype(np.float32)      # norms over 4 decades
    tab[::1000] = tab[1::1000][: len(tab[::1000])]                         # exact duplicates → ties
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    q = (base[rng.integers(0, 64, nq)] + 0.01 * rng.standard_normal((nq, d))).astype(np.float32)
    q[::3] *= -1.0                                                          # negative best scores too
    rows, scores, _ = t.recall_topk(q, k)
    orow, osc = o.recall_topk(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    # a non-finite table cannot be screened: the call falls back to the exact 64-query kernel
    tab[5, 7] = np.inf
    t.upload(tab)
    rows, scores, _ = t.recall_topk(q, k)
    orow, osc = o.recall_topk(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    t.destroy()


def test_recall_non_finite_table_with_many_queries(ctx):
    """A table with non-finite rows cannot be screened; batches beyond 64 queries used to be refused
    (PG_ERR_UNSUPPORTED) and now ride the exact fp32-MFMA scan in groups of 64 queries per launch —
    100 and 256 queries, ids / order / score bits against the oracle (NaN scores sort last)."""
    rng = np.random.default_rng(41)
    n, d, k = 70_000, 128, 150
    tab = rng.standard_normal((n, d)).astype(np.float32)
    tab[17, 3] = np.inf
    tab[4000, 100] = np.nan
    tab[69_999, 0] = -np.inf
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    assert t.screen_info()[0] == 0                       # exact scan
    for nq in (100, 256):
        q = rng.standard_normal((nq, d)).astype(np.float32)
        rows, scores, cnt = t.recall_topk(q, k)
        orow, osc = o.recall_topk(tab, q, k)
        assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc)) and cnt.tolist() == [k] * nq
    t.destroy()


def test_recall_bf16_screen_with_huge_row_norms(ctx):
    """dim 64 is screened in bf16.  Rows and queries of norm ~1e18-1e19 are finite, but their bf16 partial sums can
    reach inf - inf = NaN, which a max chain drops silently: the screen must stand aside for such
    magnitudes (threshold -inf: every row is re-scored exactly).  Half the queries are huge, half ordinary."""
    rng = np.random.default_rng(31)
    n, d, k, nq = 90_000, 64, 300, 40
    tab = rng.standard_normal((n, d)).astype(np.float32)
    tab[::7] *= np.float32(2e18)
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[::2] *= np.float32(1e18)
    rows, scores, _ = t.recall_topk(q, k)
    orow, osc = o.recall_topk(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    t.destroy()


def test_recall_int8_screen_is_exact_on_hostile_data(ctx):
    """dim 128 is screened on an int8 shadow with ONE scale for the table (max|x| / 127) and an error bound that
    uses the measured quantisation residual.  Data built against exactly that: an outlier that coarsens the scale
    (13 quantisation levels per sigma), rows of tiny magnitude, zero rows, a zero query, a query with one dominant
    component, and winners that differ by less than one quantisation step — ids, order and score bits must match
    the oracle.  An outlier large enough to make the scale useless sends the table to the bf16 shadow instead
    (relative bound, per-block row norms): same exactness."""
    rng = np.random.default_rng(23)
    n, d, k, nq = 150_000, 128, 600, 72
    tab = rng.standard_normal((n, d)).astype(np.float32) * 0.05
    tab[1234, 17] = 0.5                                                # 10 sigma: int8 stays, with a coarse step
    tab[2000:2600] *= 1e-6                                             # tiny rows
    tab[3000:3100] = 0.0                                               # zero rows
    near = (rng.standard_normal(d) * 0.05).astype(np.float32)
    tab[5000:5400] = near * (1.0 + 1e-6 * np.arange(400, dtype=np.float32)[:, None])   # sub-step differences
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = 0.0                                                         # zero query: every score is 0 → ties by row id
    q[1] = 0.0
    q[1, 17] = 1.0                                                     # one component: picks the outlier's column
    q[2] = near
    q[3] = -near
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    eb, scale, resid = t.screen_info()
    assert eb == 1 and 0.003 < scale < 0.005 and resid > 0.0
    rows, scores, _ = t.recall_topk(q, k)
    orow, osc = o.recall_topk(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    # a huge outlier: one int8 scale would quantise every ordinary row to zero → bf16 shadow
    tab[1234, 17] = 4.0e4
    t.upload(tab)
    assert t.screen_info()[0] == 2
    rows, scores, _ = t.recall_topk(q, k)
    orow, osc = o.recall_topk(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    t.destroy()


def test_recall_small_batch_4bit_screen_is_exact(ctx):
    """Batches of <= 4 queries stream a 4-bit shadow (one scale and one measured residual per row, csrc/recall_i4.hip)
    in the pilot plan's full pass.  Forced on for a small table (i4_min_rows 0, a quarter of the rows as the sample),
    with data built against it — an outlier inside a row (coarsens that row's scale), tiny rows, zero rows, winners
    that differ by far less than a 4-bit step, a zero query, one-hot and non-finite queries: ids, order and score bits
    must match the oracle, the pass must really have read the narrow shadow, and the shadow must follow uploads and
    swaps.  A table with heavy-tailed ELEMENTS (lambda above the limit) stays on the wider shadow; one with heavy-tailed
    ROW NORMS (bf16 main shadow) still uses the 4-bit one."""
    rng = np.random.default_rng(29)
    n, d, k = 400_000, 128, 200
    tab = rng.standard_normal((n, d)).astype(np.float32) * 0.05
    tab[1234, 17] = 0.5
    tab[2000:2600] *= 1e-6
    tab[3000:3100] = 0.0
    near = (rng.standard_normal(d) * 0.05).astype(np.float32)
    tab[5000:5400] = near * (1.0 + 1e-6 * np.arange(400, dtype=np.float32)[:, None])
    qs = rng.standard_normal((12, d)).astype(np.float32)
    qs[0] = 0.0
    qs[1] = 0.0
    qs[1, 17] = 1.0
    qs[2] = near
    qs[3] = -near
    qs[4, 5] = np.inf
    qs[5, 9] = np.nan
    qs[6] *= np.float32(1e-20)
    qs[7] *= np.float32(1e15)
    # (i4_max_lambda: Gaussian rows sit at lambda 1.3, above the default limit for 4 queries — lifted here so that
    # every batch size exercises the narrow shadow; the matrix-pipe screen that serves such batches by default is switched off:
    # this is the vector screen's test — squared-Euclidean recalls and tables on a bf16 main shadow have no other)
    for name, v in (("i4_min_rows", "0"), ("pilot_fraction", "0.25"), ("i4_max_lambda", "3"), ("no_screen_i4m", "1")):
        ctx.set_option(name, v)
    try:
        t = pa.Table(ctx, n, d)
        t.upload(tab)

        def check(t_, tab_, q, expect_i4=True):
            rows, scores, _ = t_.recall_topk(q, k)
            _, nbytes = ctx.last_scan_kernel()
            orow, osc = o.recall_topk(tab_, q, k)
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
            finite = bool(np.all(np.isfinite(q)))
            if finite:          # (a non-finite query overflows the pilot plan by design: the next plan answers)
                # less than one pass over the table's main shadow (int8: 128 B per row, bf16: 256) <=> the full pass was 4-bit
                assert (nbytes < n * 128 * t_.screen_info()[0]) == expect_i4, (nbytes, expect_i4)

        for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 5), (5, 6), (6, 8), (8, 11), (8, 12), (0, 4)):
            check(t, tab, qs[lo:hi])
        tab2 = tab.copy()
        tab2[100_000:160_000] = rng.standard_normal((60_000, d)).astype(np.float32) * 0.2
        t.upload(tab2[100_000:160_000], row0=100_000)
        check(t, tab2, qs[8:10])
        other = pa.Table(ctx, n, d)
        tab3 = rng.uniform(-1, 1, (n, d)).astype(np.float32)
        other.upload(tab3)
        check(other, tab3, qs[8:11])
        t.swap(other)
        check(t, tab3, qs[9:10])
        check(other, tab2, qs[10:12])
        ctx.set_option("no_screen_i4", "1")
        check(t, tab3, qs[8:9], expect_i4=False)
        ctx.set_option("no_screen_i4", "0")
        ctx.set_option("i4_max_lambda", "1.7")
        # Student-t(2.2) columns scaled into the int8 shadow's range: rows whose largest element is far above the rest
        heavy = (rng.standard_t(2.2, (n, d)) * 0.01).astype(np.float32)
        np.clip(heavy, -0.4, 0.4, out=heavy)
        other.upload(heavy)
        check(other, heavy, qs[8:9], expect_i4=False)              # elements far above their row's norm: lambda > limit
        # rows of very different magnitude (log-normal row scales): one int8 scale is useless → bf16 main shadow, but
        # every term of the 4-bit bound is relative to the row, so small batches still stream 68 B per row
        scaled = tab * np.exp(rng.standard_normal((n, 1)) * 1.5).astype(np.float32)
        other.upload(scaled)
        assert other.screen_info()[0] == 2
        check(other, scaled, qs[8:10], expect_i4=True)
        t.destroy()
        other.destroy()
        # a ragged end (rows not a multiple of the 64-row groups / 32-row blocks), winners in the last rows
        n3 = 300_011
        tab4 = rng.standard_normal((n3, d)).astype(np.float32) * 0.05
        tab4[-5:] *= 6.0
        r3 = pa.Table(ctx, n3, d)
        r3.upload(tab4)
        for lo, hi in ((8, 9), (8, 12)):
            rows, scores, _ = r3.recall_topk(qs[lo:hi], k)
            orow, osc = o.recall_topk(tab4, qs[lo:hi], k)
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
            if hi - lo <= 2:         # (four Gaussian queries are above the default lambda limit: wide shadow)
                assert ctx.last_scan_kernel()[1] < n3 * 128 * r3.screen_info()[0]      # (the 6x rows send the main shadow to bf16)
        r3.destroy()
    finally:
        ctx.set_option("i4_min_rows", str(1 << 22))
        ctx.set_option("pilot_fraction", "0")
        ctx.set_option("i4_max_lambda", "1.7")
        ctx.set_option("no_screen_i4m", "0")


def test_recall_heavy_tailed_table_stays_fast(ctx):
    """Student-t(3) rows: the largest element is ~100 x a typical row's, one int8 scale would make every row a
    suspect (every plan overflows down to the bounded-chunk one: 560 ms per recall instead of 2.4).  The table
    statistics route it to the bf16 shadow, whose bound is relative to each 32-row block's own largest norm: exact
    as ever, and the pilot plan completes — no rescans."""
    rng = np.random.default_rng(31)
    n, d, k, nq = 2_200_000, 128, 400, 96
    t = pa.Table(ctx, n, d)
    tab = np.empty((n, d), dtype=np.float32)
    for r0 in range(0, n, 200_000):
        tab[r0:r0 + 200_000] = rng.standard_t(3, (200_000, d)).astype(np.float32)
    t.upload(tab)
    assert t.screen_info()[0] == 2
    q = rng.standard_normal((nq, d)).astype(np.float32)
    before = ctx.stats().recall_rescans
    rows, scores, _ = t.recall_topk(q, k)
    assert ctx.stats().recall_rescans == before
    orow, osc = o.recall_topk(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    t.destroy()


def test_table_screen_info(ctx):
    """pg_table_screen_info: which shadow a table is screened on (int8 at dim 128, bf16 at dim 64, none when the
    rows are not finite or the width has no screen), and the int8 statistics against numpy."""
    rng = np.random.default_rng(5)
    tab = rng.uniform(-1, 1, (40_000, 128)).astype(np.float32)
    t = pa.Table(ctx, 40_000, 128)
    t.upload(tab)
    eb, scale, resid = t.screen_info()
    assert eb == 1
    s = np.float32(np.abs(tab).max()) / np.float32(127.0)
    assert abs(scale - s) <= 1e-6 * s
    X = np.clip(np.rint(tab.astype(np.float64) / float(scale)), -127, 127)
    r = np.sqrt(((tab.astype(np.float64) - float(scale) * X) ** 2).sum(axis=1)).max()
    assert r <= resid <= r * 1.01 + 1e-4                               # an upper bound, and a tight one
    tab[7, 7] = np.nan
    t.upload(tab)
    assert t.screen_info()[0] == 0
    t.destroy()
    t64 = pa.Table(ctx, 5000, 64)
    t64.fill_synthetic(o.SEED_TABLE)
    assert t64.screen_info() == (2, 0.0, 0.0)
    t64.destroy()
    t256 = pa.Table(ctx, 5000, 256)
    t256.fill_synthetic(o.SEED_TABLE)
    assert t256.screen_info()[0] == 0
    t256.destroy()


def test_recall_random_shapes_bitexact(ctx):
    """Seeded random sweep over table sizes (down to a single row, ragged tails), widths, K and batch sizes:
    every kernel variant and plan of the recall (seed / screened sample / full pass / geometric chunks)
    against the oracle, bit for bit."""
    rng = np.random.default_rng(2026)
    shapes = [(1, 64, 1, 1), (31, 128, 40, 3), (33, 64, 33, 70), (100, 128, 5, 256), (513, 64, 512, 130)]
    for _ in range(10):
        n = int(rng.integers(600, 260_000))
        shapes.append((n, int(rng.choice([64, 128])), int(rng.integers(1, min(n, 3000))), int(rng.integers(1, 257))))
    shapes.append((2_100_001, 128, 700, 200))                # pilot plan on a ragged table
    for n, d, k, nq in shapes:
        off = int(rng.integers(0, 1000))
        t = pa.Table(ctx, n, d, row_offset=off)
        t.fill_synthetic(o.SEED_TABLE)
        tab = o.synth_rows(o.SEED_TABLE, off, n, d)
        q = o.synth_rows(o.SEED_QUERY, int(rng.integers(0, 500)), nq, d)
        rows, scores, cnt = t.recall_topk(q, k)
        m = min(k, n)
        orow, osc = o.recall_topk(tab, q, m, row_offset=off)
        assert cnt.tolist() == [m] * nq, (n, d, k, nq)
        assert np.array_equal(rows[:, :m], orow), (n, d, k, nq)
        assert np.array_equal(bits(scores[:, :m]), bits(osc)), (n, d, k, nq)
        assert np.all(rows[:, m:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isneginf(scores[:, m:]))
        t.destroy()


@pytest.mark.parametrize("n,d,k,nq", [
    (1000, 64, 200, 1),
    (140_000, 128, 500, 32),
    (300_017, 128, 1000, 50),     # two 32-query column blocks, ragged row count
    (250_000, 64, 300, 130),      # three groups of <= 64 queries
    (2_300_000, 128, 2000, 3),    # pilot plan on the exact scan
])
def test_recall_l2_matches_oracle_bitexact(ctx, n, d, k, nq):
    """HologresVectorRecallV2 (service/recall/hologres_vector_recall_v2.go:23): the k rows of smallest squared Euclidean
    distance, ascending, distance as the score.  Rows of different norms (so that the order differs from the inner
    product's), duplicates (ties by row id), a query equal to a row (distance ~ 0): ids, order and distance bits = oracle."""
    rng = np.random.default_rng(n + nq)
    tab = rng.standard_normal((n, d)).astype(np.float32) * rng.uniform(0.3, 2.0, (n, 1)).astype(np.float32)
    tab[100:140] = tab[100]
    q = rng.standard_normal((nq, d)).astype(np.float32)
    q[0] = tab[7]
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    rows, dist, cnt = t.recall_topk_l2(q, k)
    orow, od = o.recall_topk_l2(tab, q, k)
    m = min(n, k)
    assert cnt.tolist() == [m] * nq
    assert np.array_equal(rows[:, :m], orow) and np.array_equal(bits(dist[:, :m]), bits(od))
    assert np.all(np.diff(dist[:, :m].astype(np.float64), axis=1) >= 0)
    irow, _, _ = t.recall_topk(q[:2], k)
    assert not np.array_equal(irow[:, :m], rows[:2, :m])        # not the inner product's order
    # the norms follow an upload
    tab[5] = q[min(1, nq - 1)]
    t.upload(tab[5:6], 5)
    rows, dist, _ = t.recall_topk_l2(q[:2], k)
    orow, od = o.recall_topk_l2(tab, q[:2], k)
    assert np.array_equal(rows[:, :m], orow) and np.array_equal(bits(dist[:, :m]), bits(od))
    t.destroy()


def test_recall_l2_screened_pass_on_rows_of_equal_norm(ctx):
    """Squared-Euclidean recall of a dim-128 table whose rows have (nearly) one norm: the pass streams the int8 shadow with
    per-block integer cutoffs (csrc/recall.hip, screen_thr8_l2_kernel) and re-scores the suspects exactly — ids, order
    and distance bits still the oracle's, no plan fails (inactive query columns must not produce suspects), and the pass
    reads a quarter of the fp32 bytes.  Rows of very different norms: the per-row form of the test, on the same shadow."""
    n, d, k = 1_200_000, 128, 400
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    tab[500:532] = tab[500]                                     # a block of duplicates: ties by row id
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    assert t.screen_info()[0] == 1
    for nq in (1, 3, 40, 128, 200):
        q = o.synth_rows(o.SEED_QUERY, 50, nq, d) * np.float32(1.7)
        q[0] = tab[500]
        before = ctx.stats().recall_rescans
        rows, dist, cnt = t.recall_topk_l2(q, k)
        assert ctx.stats().recall_rescans == before, nq
        _, nbytes = ctx.last_scan_kernel()
        assert nbytes < n * d * 2, (nq, nbytes)                # the int8 shadow (+ samples), not the fp32 rows
        sel = sorted(set([0, nq - 1, nq // 2]))
        orow, od = o.recall_topk_l2(tab, q[sel], k)
        assert np.array_equal(rows[sel], orow) and np.array_equal(bits(dist[sel]), bits(od)), nq
    assert rows[0, :32].tolist() == list(range(500, 532))
    # norms spread over a factor of four (slack ~ 8 score spreads): one cutoff per block would give too much away — the pass
    # tests every row against its own norm instead (still the int8 shadow); same answers, and the same from the exact scan
    scaled = tab * np.linspace(0.5, 2.0, n, dtype=np.float32)[::-1, None].copy()
    rng = np.random.default_rng(3)
    scaled = scaled[rng.permutation(n)]
    t.upload(scaled)
    for nq in (5, 70, 128):
        q = o.synth_rows(o.SEED_QUERY, 9, nq, d) * np.float32(1.3)
        before = ctx.stats().recall_rescans
        rows, dist, _ = t.recall_topk_l2(q, k)
        assert ctx.stats().recall_rescans == before and ctx.last_scan_kernel()[1] < n * d * 2, nq
        sel = sorted(set([0, nq - 1, nq // 2]))
        orow, od = o.recall_topk_l2(scaled, q[sel], k)
        assert np.array_equal(rows[sel], orow) and np.array_equal(bits(dist[sel]), bits(od)), nq
    ctx.set_option("l2_exact", "1")
    rows2, dist2, _ = t.recall_topk_l2(q[:9], k)
    ctx.set_option("l2_exact", "0")
    assert ctx.last_scan_kernel()[1] >= n * d * 4
    assert np.array_equal(rows2, rows[:9]) and np.array_equal(bits(dist2), bits(dist[:9]))
    t.destroy()


def test_recall_l2_small_batches_on_the_4bit_shadow(ctx):
    """Squared-Euclidean recalls of 1-4 queries stream the 4-bit shadow in their full pass (csrc/recall_i4.hip, L2 form of the
    per-row test: 2 x the inner-product bound - |x|^2 - |q|^2 against the threshold) — forced on for a small table: ids,
    order and distance bits = oracle on rows of one norm and of mixed norms, the pass reads less than the int8 shadow, and
    the same answers come from the int8 pass."""
    rng = np.random.default_rng(61)
    n, d, k = 500_000, 128, 300
    for name, v in (("i4_min_rows", "0"), ("pilot_fraction", "0.25"), ("i4_max_lambda", "3")):
        ctx.set_option(name, v)
    try:
        base = o.synth_rows(o.SEED_TABLE, 0, n, d)
        mixed = (base * rng.uniform(0.4, 1.8, (n, 1)).astype(np.float32)).astype(np.float32)
        t = pa.Table(ctx, n, d)
        for tab in (base, mixed):
            t.upload(tab)
            for nq in (1, 2, 4):
                q = (o.synth_rows(o.SEED_QUERY, 70 + nq, nq, d) * np.float32(1.2)).astype(np.float32)
                before = ctx.stats().recall_rescans
                rows, dist, _ = t.recall_topk_l2(q, k)
                assert ctx.stats().recall_rescans == before
                assert ctx.last_scan_kernel()[1] < n * d, "the full pass did not stream the 4-bit shadow"
                orow, od = o.recall_topk_l2(tab, q, k)
                assert np.array_equal(rows, orow) and np.array_equal(bits(dist), bits(od)), nq
                ctx.set_option("no_screen_i4", "1")
                rows8, dist8, _ = t.recall_topk_l2(q, k)
                ctx.set_option("no_screen_i4", "0")
                assert np.array_equal(rows8, rows) and np.array_equal(bits(dist8), bits(dist))
        t.destroy()
    finally:
        ctx.set_option("i4_min_rows", str(1 << 22))
        ctx.set_option("pilot_fraction", "0")
        ctx.set_option("i4_max_lambda", "1.7")


def test_recall_with_a_where_clause_matches_the_oracle_on_the_admitted_rows(ctx):
    """A Hologres vector recall with its WhereClause (hologres_vector_recall.go:23,49-62 / _v2.go:23: "FROM table WHERE …
    ORDER BY distance LIMIT n"), `column OP constant` over an integer column keyed by item row: only rows that pass are
    candidates.  Against the oracle run on the admitted rows alone (same row ids, same tie order) for selectivities from one
    row in two to fewer rows than K, both metrics, 1 to 200 queries, int32 and int64 columns, every operator."""
    rng = np.random.default_rng(71)
    n, d, k = 900_000, 128, 500
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d) * rng.uniform(0.7, 1.3, (n, 1)).astype(np.float32)
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    feats = pa.Features(ctx, n)
    ts32 = rng.integers(0, 1_000_000, n).astype(np.int32)
    ts64 = (rng.integers(0, 1_000_000, n).astype(np.int64) + (1 << 40))
    feats.set_column("create_time", pa.F_I32, ts32)
    feats.set_column("stamp64", pa.F_I64, ts64)
    ops = {">": np.greater, ">=": np.greater_equal, "<": np.less, "<=": np.less_equal, "==": np.equal, "!=": np.not_equal}

    def check(col, vals, op, value, nq, l2):
        mask = ops[op](vals, value)
        idx = np.nonzero(mask)[0]
        q = (o.synth_rows(o.SEED_QUERY, 11 * nq, nq, d) * np.float32(1.1)).astype(np.float32)
        rows, sc, cnt = t.recall_topk_where(feats, col, op, value, q, k, l2=l2)
        m = min(k, idx.size)
        assert cnt.tolist() == [m] * nq, (op, value, cnt[:4], m)
        sel = sorted(set([0, nq - 1, nq // 2]))
        if m:
            f = o.recall_topk_l2 if l2 else o.recall_topk
            orow, osc = f(tab[idx], q[sel], k)
            assert np.array_equal(rows[sel][:, :m], idx[orow.astype(np.int64)].astype(np.uint64)), (op, value, nq, l2)
            assert np.array_equal(bits(sc[sel][:, :m]), bits(osc)), (op, value, nq, l2)
        assert np.all(rows[:, m:] == np.uint64(0xFFFFFFFFFFFFFFFF))

    check("create_time", ts32, ">", 500_000, 40, False)          # one row in two
    check("create_time", ts32, ">=", 900_000, 200, False)        # one in ten, 200 queries (hit records + filter)
    check("create_time", ts32, "<", 10_000, 3, False)            # one in a hundred, a small batch (4-bit pass + filter)
    check("create_time", ts32, "<=", 300, 5, False)              # ~ 270 rows: fewer than K
    check("create_time", ts32, "==", int(ts32[12345]), 2, False)  # a handful of rows
    check("create_time", ts32, "!=", int(ts32[0]), 17, False)
    check("stamp64", ts64, ">", (1 << 40) + 700_000, 9, False)
    check("create_time", ts32, ">", 500_000, 40, True)           # squared Euclidean
    check("create_time", ts32, ">=", 990_000, 130, True)
    check("stamp64", ts64, "<", (1 << 40) + 200, 1, True)        # fewer than K, one query
    check("create_time", ts32, "<", 0, 6, False)                 # nothing passes
    check("create_time", ts32, ">", 2_000_000, 1, True)
    # (filters that admit at most an eighth of the table are served from a compact copy of the admitted rows; the same
    #  answers with the predicate applied in place)
    ctx.set_option("where_compact_max_rows", 0)
    check("create_time", ts32, ">=", 900_000, 200, False)
    check("create_time", ts32, "<", 10_000, 3, False)
    check("create_time", ts32, "<=", 300, 5, False)
    check("create_time", ts32, ">=", 990_000, 130, True)
    # a handful of admitted rows, 256 queries: thresholds stay open through every chunk of the last-resort plan (it must not
    # put such chunks on the screened scan: every row would be a suspect of every query)
    check("create_time", ts32, "==", int(ts32[12345]), 256, False)
    check("create_time", ts32, "==", int(ts32[777]), 200, True)
    ctx.set_option("where_compact_max_rows", 8 << 20)
    with pytest.raises(pa._lib.PgError):
        feats.set_column("price", pa.F_F32, np.zeros(n, np.float32))
        t.recall_topk_where(feats, "price", ">", 1, tab[:1], k)
    feats.destroy()
    t.destroy()


def test_recall_hit_records_on_a_table_in_ascending_score_order(ctx):
    """The last-resort plan's case — every later row beats every earlier one, for every query — at more than 128 queries
    (hit records) and a small K (a small spill pool): each bounded chunk makes every row a suspect of every query, and the
    per-wave record regions must hold that for the uneven split of a SIMD's blocks between its two waves."""
    rng = np.random.default_rng(3)
    d = 128
    for n, k, nq in ((1_200_000, 10, 256), (1_000_000, 1, 200), (1_100_000, 300, 129)):
        v = rng.standard_normal(d).astype(np.float32)
        v /= np.linalg.norm(v)
        scale = np.linspace(0.2, 1.0, n, dtype=np.float32)[:, None]
        tab = (v[None, :] * scale + 0.002 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
        q = (v[None, :] + 0.05 * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
        t = pa.Table(ctx, n, d)
        t.upload(tab)
        rows, sc, cnt = t.recall_topk(q, k)
        sel = [0, nq // 2, nq - 1]
        orow, osc = o.recall_topk(tab, q[sel], k)
        assert np.array_equal(rows[sel], orow) and np.array_equal(bits(sc[sel]), bits(osc)), (n, k, nq)
        assert cnt.tolist() == [k] * nq
        t.destroy()


def test_filtered_view_recalls_report_the_source_rows(ctx):
    """pg_table_view_create: the rows a WhereClause admits (constant fixed when the recall is built, hologres_vector_recall.go:
    56-61) as a table of their own.  Every recall flavour over the view — inner product and squared Euclidean, 1 to 200
    queries (4-bit pass, screened pass, hit records), the coalescer's per-request calls from 96 threads — answers with the
    SOURCE table's row ids, exactly the oracle's top-K over the admitted rows (ties by source row).  Views serve recalls only."""
    import threading
    rng = np.random.default_rng(5)
    n, d, k = 600_000, 128, 400
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d) * rng.uniform(0.8, 1.25, (n, 1)).astype(np.float32)
    tab[1000:1040] = tab[2000:2040]                                       # duplicates: ties by source row
    off = 1 << 20
    t = pa.Table(ctx, n, d, row_offset=off)
    t.upload(tab)
    feats = pa.Features(ctx, n)
    cat = rng.integers(0, 10, n).astype(np.int32)
    cat[1000:1040] = 3
    cat[2000:2040] = 3
    feats.set_column("cat", pa.F_I32, cat)
    feats.set_column("stamp", pa.F_I64, np.arange(n, dtype=np.int64) + (1 << 41))
    for col, op, value, mask in (("cat", "==", 3, cat == 3), ("stamp", ">=", (1 << 41) + n - 300, np.arange(n) >= n - 300),
                                 ("cat", "!=", 0, cat != 0)):
        idx = np.nonzero(mask)[0]
        v = t.view(feats, col, op, value)
        assert v.rows == idx.size
        m = min(k, idx.size)
        for nq, l2 in ((1, False), (3, True), (40, False), (130, True), (200, False)):
            q = (o.synth_rows(o.SEED_QUERY, 17 * nq, nq, d) * np.float32(1.05)).astype(np.float32)
            rows, sc, cnt = (v.recall_topk_l2 if l2 else v.recall_topk)(q, k)
            sel = sorted(set([0, nq // 2, nq - 1]))
            orow, osc = (o.recall_topk_l2 if l2 else o.recall_topk)(tab[idx], q[sel], k)
            assert cnt.tolist() == [m] * nq
            assert np.array_equal(rows[sel][:, :m], (idx[orow[:, :m].astype(np.int64)] + off).astype(np.uint64)), (col, nq, l2)
            assert np.array_equal(bits(sc[sel][:, :m]), bits(osc[:, :m])), (col, nq, l2)
            assert np.all(rows[:, m:] == np.uint64(0xFFFFFFFFFFFFFFFF))
        if col == "cat" and op == "==":
            # the same answers as the per-call form, and through a coalescer over the view
            q = o.synth_rows(o.SEED_QUERY, 900, 96, d)
            wr, ws, _ = t.recall_topk_where(feats, col, op, value, q, k)
            vr, vs, _ = v.recall_topk(q, k)
            assert np.array_equal(wr, vr) and np.array_equal(bits(ws), bits(vs))
            co = pa.Coalescer(ctx, v, k, algos=[], max_wait_us=2000)
            got = [None] * 96
            th = [threading.Thread(target=lambda i=i: got.__setitem__(i, co.recall_l2(q[i]) if i % 2 else co.recall(q[i]))) for i in range(96)]
            [x.start() for x in th]
            [x.join() for x in th]
            lr, ls, _ = v.recall_topk_l2(q, k)
            for i in range(96):
                want = (lr[i], ls[i]) if i % 2 else (vr[i], vs[i])
                assert np.array_equal(got[i][0], want[0]) and np.array_equal(bits(got[i][1]), bits(want[1])), i
            with pytest.raises(pa._lib.PgError):
                co.i2i_recall(5)                                         # trigger rows are rows of the source
            co.destroy()
        v.destroy()
    with pytest.raises(pa._lib.PgError):
        t.view(feats, "cat", ">", 100)                                    # nothing passes
    feats.destroy()
    t.destroy()


def test_recall_follows_table_updates(ctx):
    """The screen streams a quantised shadow (int8 here) of the table that is built lazily; uploads, synthetic fills and
    hot swaps must invalidate / carry it — every recall answers for the rows the table holds now."""
    n, d, k = 60000, 128, 300
    a = o.synth_rows(o.SEED_TABLE, 0, n, d)
    b = o.synth_rows(o.SEED_TABLE, 10 * n, n, d)
    q = o.synth_rows(o.SEED_QUERY, 0, 40, d)
    ta, tb = pa.Table(ctx, n, d), pa.Table(ctx, n, d)
    ta.upload(a)
    tb.upload(b)

    def check(t, tab):
        rows, scores, _ = t.recall_topk(q, k)
        orow, osc = o.recall_topk(tab, q, k)
        assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))

    check(ta, a)                                   # builds ta's shadow
    a2 = a.copy()
    a2[1000:3000] = b[1000:3000] * 3.0             # partial upload: new rows, new max norm
    ta.upload(a2[1000:3000], row0=1000)
    check(ta, a2)
    check(tb, b)
    ta.swap(tb)                                    # both shadows valid: they travel with the rows
    check(ta, b)
    check(tb, a2)
    ta.fill_synthetic(o.SEED_TABLE)                # refill in place
    check(ta, o.synth_rows(o.SEED_TABLE, 0, n, d))
    ta.destroy()
    tb.destroy()


def test_recall_adversarial_order(ctx):
    """Scores ascending with the row index defeat a running threshold (every row beats everything seen
    before it).  The pilot plan samples the whole table and is immune; with the pilot disabled the
    growing-chunk plan overflows its candidate lists and the bounded-chunk plan must take over."""
    n, d, k = 2_200_000, 64, 100
    tab = np.zeros((n, d), dtype=np.float32)
    tab[:, 3] = np.arange(n, dtype=np.float32)
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    q = np.zeros((1, d), dtype=np.float32)
    q[0, 3] = 1.0
    before = ctx.stats().recall_rescans
    rows, scores, _ = t.recall_topk(q, k)
    assert rows[0].tolist() == list(range(n - 1, n - 1 - k, -1))
    assert ctx.stats().recall_rescans == before
    ctx.set_option("no_pilot", 1)
    try:
        rows, scores, _ = t.recall_topk(q, k)
    finally:
        ctx.set_option("no_pilot", 0)
    assert rows[0].tolist() == list(range(n - 1, n - 1 - k, -1))
    assert ctx.stats().recall_rescans == before + 1
    t.destroy()


def test_recall_threshold_refinement_any_row_order(ctx):
    """The pilot plan raises its threshold after the first quarter of the full pass to the k2-th best candidate found
    so far (csrc/recall.hip).  A contiguous prefix is not a random sample: forced on for a small table, with the rows
    ordered so that the prefix holds none, all, or a disproportionate share of the best rows — results must be exact
    for every order, and for the random order no plan may fail."""
    rng = np.random.default_rng(41)
    n, d, k, nq = 400_000, 64, 500, 40
    base = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    key = base @ q[0]                                          # order by the first query's score
    asc = np.argsort(key)
    orders = {
        "random": np.arange(n),
        "ascending": asc,                                      # the prefix holds the worst rows
        "descending": asc[::-1],                               # ... all of the best
        "clustered": np.concatenate([asc[-3000:][::2], asc[:n - 3000], asc[-3000:][1::2]]),   # ... half of the best
    }
    for name, v in (("refine_min_rows", "0"), ("pilot_fraction", "0.125")):
        ctx.set_option(name, v)
    try:
        t = pa.Table(ctx, n, d)
        for name, perm in orders.items():
            tab = np.ascontiguousarray(base[perm])
            t.upload(tab)
            before = ctx.stats().recall_rescans
            rows, scores, _ = t.recall_topk(q, k)
            launches = ctx.last_scan_launches() if hasattr(ctx, "last_scan_launches") else None
            orow, osc = o.recall_topk(tab, q, k)
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc)), name
            if name == "random":
                assert ctx.stats().recall_rescans == before, "the refined threshold failed on randomly ordered rows"
            # and the same answers without the refinement
            ctx.set_option("no_refine", "1")
            rows2, scores2, _ = t.recall_topk(q, k)
            ctx.set_option("no_refine", "0")
            assert np.array_equal(rows2, orow) and np.array_equal(bits(scores2), bits(osc)), name
        t.destroy()
    finally:
        ctx.set_option("refine_min_rows", str(1 << 24))
        ctx.set_option("pilot_fraction", "0")


def test_recall_hit_records_with_every_lane_hit(ctx):
    """The hit-record path of the > 128-query int8 screen (csrc/recall.hip, kRecBytes) under data that fills whole tiles
    with suspects: zero queries (threshold 0: every row of every tile is a suspect of that query — more hit lanes per
    tile than the wave's LDS stage holds, regions and spill pool overflow, the plan falls back), blocks of identical rows
    (ties by row id inside one tile), a ragged row count (the last block's rows past the end must not be emitted).
    Ids, order and score bits must match the oracle for every query."""
    rng = np.random.default_rng(83)
    n, d, k, nq = 300_011, 128, 700, 200
    tab = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    tab /= np.linalg.norm(tab, axis=1, keepdims=True)
    tab[1000:1064] = tab[1000]                                  # 64 identical rows: two whole tiles of ties
    tab[n - 11:] = tab[1000] * 1.5                              # the best rows sit in the ragged last block
    q = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    q[0] = 0.0
    q[77] = 0.0
    q[150] = tab[1000]
    q[151] = -tab[1000]
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    assert t.screen_info()[0] == 1
    rows, scores, cnt = t.recall_topk(q, k)
    qs = [0, 77, 150, 151, 3, 199]
    orow, osc = o.recall_topk(tab, q[qs], k)
    assert np.array_equal(rows[qs], orow) and np.array_equal(bits(scores[qs]), bits(osc))
    assert rows.max() < n
    t.destroy()


def test_recall_hit_records_spill_pool_on_a_table_whose_best_rows_sit_together(ctx, capfd):
    """More than 128 queries on the int8 shadow: the scan parks hit records (a lane's 16 accumulators + a tag) in a region
    per wave and screen_decode_kernel turns them into the suspect lists (csrc/recall.hip, kRecBytes).  Rows of three times
    the norm packed into a few thousand rows: every query's best rows — and nearly all suspects — fall into a handful of
    wave regions, which overflow into the shared spill pool.  Answers must be exact, without a fallback to another plan."""
    rng = np.random.default_rng(77)
    n, d, k, nq = 3_000_000, 128, 2000, 200
    tab = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    tab /= np.linalg.norm(tab, axis=1, keepdims=True)
    hot = slice(1_500_000, 1_506_000)                          # 6 000 rows: a handful of 32-row-block runs
    tab[hot] *= 3.0
    q = rng.uniform(-1, 1, (nq, d)).astype(np.float32)
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    ctx.set_option("debug_scan", "1")
    try:
        before = ctx.stats().recall_rescans
        rows, scores, _ = t.recall_topk(q, k)
        assert ctx.stats().recall_rescans == before, "the spill pool should have absorbed the skew without a re-plan"
    finally:
        ctx.set_option("debug_scan", "0")
    err = capfd.readouterr().err
    spilled = [int(l.rsplit(",", 1)[1].split()[0]) for l in err.splitlines() if "in the spill pool" in l]
    assert spilled and max(spilled) > 0, "the skewed table did not exercise the spill pool:\n" + err[-2000:]
    qs = rng.choice(nq, 6, replace=False)
    orow, osc = o.recall_topk(tab, q[qs], k)
    assert np.array_equal(rows[qs], orow) and np.array_equal(bits(scores[qs]), bits(osc))
    assert np.mean((rows >= hot.start) & (rows < hot.stop)) > 0.25     # a large part of every answer comes from the hot rows
    t.destroy()


def test_recall_refinement_gives_up_on_a_table_with_an_unrepresentative_head(ctx):
    """Rows of three times the norm in the first fifth of the table: for every query the head holds all of the best
    rows, the threshold raised after the first quarter is far above the true K-th score, the verification rejects every
    query and the batch is answered by the next plan — exactly.  After two such batches the table's recalls stop
    refining (pg_table::prefix_failures): the third recall needs no re-run."""
    rng = np.random.default_rng(43)
    n, d, k, nq = 400_000, 128, 300, 24
    tab = rng.standard_normal((n, d)).astype(np.float32)
    tab[: n // 5] *= np.float32(3.0)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    for name, v in (("refine_min_rows", "0"), ("pilot_fraction", "0.125")):
        ctx.set_option(name, v)
    try:
        t = pa.Table(ctx, n, d)
        t.upload(tab)
        orow, osc = o.recall_topk(tab, q, k)
        rescans = []
        for _ in range(4):
            before = ctx.stats().recall_rescans
            rows, scores, _ = t.recall_topk(q, k)
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
            rescans.append(ctx.stats().recall_rescans - before)
        assert rescans[0] >= 1 and rescans[1] >= 1, rescans        # the raised threshold was rejected ...
        assert rescans[2] == 0 and rescans[3] == 0, rescans        # ... and the table stopped being refined
        t.destroy()
    finally:
        ctx.set_option("refine_min_rows", str(1 << 24))
        ctx.set_option("pilot_fraction", "0")


def test_recall_all_ties_falls_through_every_plan(ctx):
    """A table of identical rows: every row ties with any threshold, so the pilot pass and the growing
    chunks both overflow and the bounded-chunk plan answers; ties resolve to the lowest rows
    (sort.go SliceStable semantics of the recall order)."""
    n, d, k = 2_200_000, 64, 100
    tab = np.zeros((n, d), dtype=np.float32)
    tab[:, 5] = 1.0
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    q = np.zeros((2, d), dtype=np.float32)
    q[:, 5] = [1.0, -2.0]
    before = ctx.stats().recall_rescans
    rows, scores, _ = t.recall_topk(q, k)
    assert rows[0].tolist() == list(range(k)) and rows[1].tolist() == list(range(k))
    assert np.all(scores[0] == 1.0) and np.all(scores[1] == -2.0)
    # (the screened pilot pass overflows, the job moves to the exact scan and starts over: its pilot pass and growing chunks
    #  overflow too — every row ties — and the bounded chunks answer)
    assert before + 2 <= ctx.stats().recall_rescans <= before + 3
    t.destroy()


def test_recall_rows_crowded_within_the_screens_error_move_to_the_exact_scan(ctx):
    """Rows nearly collinear, queries along them: a per cent of the table lies within the int8 margin of every query's K-th
    score, the screened plan's hit-record areas overflow.  Round 6: the areas grow with such a table (pg_table::rec_scale) and the
    screened pass holds from then on; with the growth switched off (max_rec_scale 1, the round-3 behaviour) the job finishes on
    the exact scan (not chunk by chunk through the same crowd), and after two such batches the table's recalls start there; an
    upload resets either.  Answers exact throughout."""
    rng = np.random.default_rng(3)
    n, d, k, nq = 3_000_000, 128, 2000, 256
    v = rng.standard_normal(d).astype(np.float32)
    v /= np.linalg.norm(v)
    tab = (v[None] * rng.uniform(0.2, 1.0, (n, 1)).astype(np.float32) + 0.002 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    q = (v[None] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    orow, osc = o.recall_topk(tab, q[:2], k)

    def four_batches():
        plans, grown, fell = [], [], []
        for _ in range(4):
            s0 = ctx.stats()
            rows, sc, cnt = t.recall_topk(q, k)
            s1 = ctx.stats()
            plans.append(s1.recall_rescans - s0.recall_rescans)
            grown.append(s1.recall_record_growths - s0.recall_record_growths)
            fell.append(s1.recall_screen_overflows - s0.recall_screen_overflows)
            assert np.array_equal(rows[:2], orow) and np.array_equal(bits(sc[:2]), bits(osc)) and cnt.tolist() == [k] * nq
        return plans, grown, fell
    ctx.set_option("max_rec_scale", 1)
    try:
        plans, grown, fell = four_batches()
        assert plans[0] >= 1 and plans[1] >= 1 and plans[2] == 0 and plans[3] == 0 and sum(grown) == 0 and fell[0] >= 1, (plans, grown, fell)
        t.upload(tab[:1000], 0)                                         # new contents: the screen gets its chance again
        before = ctx.stats().recall_rescans
        t.recall_topk(q, k)
        assert ctx.stats().recall_rescans > before
    finally:
        ctx.set_option("max_rec_scale", 16)
    t.upload(tab[:1000], 0)
    plans, grown, fell = four_batches()
    assert grown[0] >= 1 and sum(grown[1:]) == 0 and sum(fell) == 0 and plans[1:] == [0, 0, 0], (plans, grown, fell)
    assert ctx.last_scan_kernel()[1] <= n * 128 * 1.2                   # ... and the pass streams the int8 shadow, not the fp32 rows
    t.destroy()


def test_topk_merge_matches_oracle(ctx):
    """The multi-GPU exchange step: G per-shard lists → global top-K (here on one device)."""
    n, d, k, G, nq = 80000, 64, 1000, 4, 3
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    q = o.synth_rows(o.SEED_QUERY, 0, nq, d)
    per = n // G
    lists_r = np.zeros((nq, G, k), dtype=np.uint64)
    lists_s = np.zeros((nq, G, k), dtype=np.float32)
    for g in range(G):
        t = pa.Table(ctx, per, d, row_offset=g * per)
        t.upload(tab[g * per:(g + 1) * per])
        r, s, _ = t.recall_topk(q, k)
        lists_r[:, g], lists_s[:, g] = r, s
        t.destroy()
    d_r, d_s = ctx.to_device(lists_r), ctx.to_device(lists_s)
    d_or, d_os = ctx.malloc(nq * k * 8), ctx.malloc(nq * k * 4)
    pa._lib.check(ctx.L.pg_topk_merge_dev(ctx.h, d_r, d_s, nq, G, k, k, d_or, d_os))
    out_r, out_s = np.zeros((nq, k), np.uint64), np.zeros((nq, k), np.float32)
    ctx.d2h(out_r, d_or)
    ctx.d2h(out_s, d_os)
    g_rows, g_scores = o.recall_topk(tab, q, k)
    assert np.array_equal(out_r, g_rows) and np.array_equal(bits(out_s), bits(g_scores))
    for p in (d_r, d_s, d_or, d_os):
        ctx.free(p)


# ---------------------------------------------------------------------------------------------
# rank
# ---------------------------------------------------------------------------------------------
def _dnn3_case(ctx, n_rows=30000):
    t = pa.Table(ctx, n_rows, 128)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 0, n_rows, 128)
    w = o.Dnn3Weights()
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    return t, tab, w, blob


def test_rank_dnn3_f32_parity(ctx):
    """PG_PREC_F32: pre-activations are bit-defined chains; only expf may differ (≤ 1 ulp of the
    fp32 sigmoid) → tolerance 2e-7 absolute, far inside the 1e-5 the north star allows."""
    t, tab, w, blob = _dnn3_case(ctx)
    rng = np.random.default_rng(1)
    sizes = [5000, 1, 0, 333, 128, 129]                      # ragged, empty, tile-boundary requests
    users = o.synth_rows(o.SEED_QUERY, 0, len(sizes), 128)
    cands = [rng.integers(0, 30000, s).astype(np.uint32) for s in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, blob)
    got = m.rank_dnn3(t, users, np.concatenate(cands), off)
    ref = np.concatenate([o.dnn3_forward(w, 0, users[r], tab[cands[r]]) for r in range(len(sizes))])
    assert got.shape == ref.shape
    assert np.max(np.abs(got.astype(np.float64) - ref)) <= 2e-7
    m.destroy()
    t.destroy()


def test_rank_dnn3_bf16_parity(ctx):
    """PG_PREC_BF16: the oracle mirrors every rounding point; the MFMA's fp32 accumulation order is
    unspecified, so scores are compared at the north star's 1e-5 (observed max ≈ 1e-6)."""
    t, tab, w, blob = _dnn3_case(ctx)
    rng = np.random.default_rng(2)
    sizes = [5000, 700]
    users = o.synth_rows(o.SEED_QUERY, 5, 2, 128)
    cands = [rng.integers(0, 30000, s).astype(np.uint32) for s in sizes]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, blob)
    got = m.rank_dnn3(t, users, np.concatenate(cands), off)
    ref = np.concatenate([o.dnn3_forward(w, 1, users[r], tab[cands[r]]) for r in range(2)])
    diff = np.abs(got.astype(np.float64) - ref)
    assert diff.max() <= 1e-5
    # and the bf16 model stays close to the fp32 model (sanity of the rounding points)
    ref32 = np.concatenate([o.dnn3_forward(w, 0, users[r], tab[cands[r]]) for r in range(2)])
    assert np.max(np.abs(got - ref32)) < 5e-3
    m.destroy()
    t.destroy()


def test_rank_dnn3_rejects_bad_input(ctx):
    t, tab, w, blob = _dnn3_case(ctx, 1000)
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, blob)
    users = o.synth_rows(o.SEED_QUERY, 0, 1, 128)
    with pytest.raises(pa._lib.PgError):                      # row outside the table
        m.rank_dnn3(t, users, np.array([5, 1000], np.uint32), [0, 2])
    with pytest.raises(pa._lib.PgError):                      # wrong blob length
        pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, blob[:-4])
    assert m.rank_dnn3(t, users, np.zeros(0, np.uint32), [0, 0]).shape == (0,)
    m.destroy()
    t.destroy()


@pytest.mark.parametrize("prec,tol", [(0, 3e-7), (1, 1e-5)])
def test_rank_fm_twotower_parity(ctx, prec, tol):
    fw = o.Fm2tWeights(vocab=3000)
    m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, prec, pa.pack_fm2t(fw))
    rng = np.random.default_rng(4)
    sizes = [5000, 77, 1]
    users = o.synth_rows(o.SEED_QUERY, 9, 3, 128)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    ufids = rng.integers(0, 3000, (3, 8)).astype(np.int32)
    ifids = rng.integers(0, 3000, (int(off[-1]), 8)).astype(np.int32)
    got = m.rank_fm2t(users, ufids, ifids, off)
    ref = np.concatenate([o.fm2t_forward(fw, prec, users[r], ufids[r], ifids[off[r]:off[r + 1]])
                          for r in range(3)])
    assert np.max(np.abs(got.astype(np.float64) - ref)) <= tol
    m.destroy()


# ---------------------------------------------------------------------------------------------
# score fusion
# ---------------------------------------------------------------------------------------------
def test_expr_reference_known_answers_on_device(ctx, golden):
    for c in golden["expr"]:
        e = pa.Expr(c["expr"])
        vals = {**c["algo_scores"], **c["properties"]}
        v = np.array([[vals[n]] for n in e.var_names], dtype=np.float64)
        got = e.eval(ctx, v)[0]
        it = o.OracleItem("x")
        for k_, v_ in vals.items():
            it.add_algo_score(k_, v_)
        want = o.expr_eval(o.expr_parse(c["expr"]), it.float_expr_data)
        if "expect" in c:
            assert got == c["expect"], c["ref"]
        else:                                                 # `^` = pow: device pow within 2 ulp of libm
            assert abs(got - want) <= 4e-16 * abs(want), c["ref"]
        e.free()


def test_expr_matches_oracle_on_5000_items(ctx):
    rng = np.random.default_rng(6)
    n = 5000
    src = "(${ctr}+2*${cvr})*${price}^0.1 + ${boost}#0.5 - ${n}%7/3"
    e = pa.Expr(src)
    cols = {"ctr": rng.random(n), "cvr": rng.random(n) * 0.1, "price": rng.random(n) * 100 + 1,
            "boost": np.where(rng.random(n) < 0.5, 0.0, rng.random(n)), "n": np.floor(rng.random(n) * 1000)}
    v = np.stack([cols[name] for name in e.var_names])
    got = e.eval(ctx, v)
    ast = o.expr_parse(src)
    want = np.array([o.expr_eval(ast, lambda name, i=i: cols[name][i]) for i in range(n)])
    # device pow() is within 2 ulp of libm; the subtraction amplifies that relative to the result,
    # so the bound is absolute against the operands' magnitude (~5)
    assert np.max(np.abs(got - want)) <= 5e-15 * 5
    # division by zero: the reference panics; here the call fails and reports it
    z = pa.Expr("1/${x}")
    with pytest.raises(pa._lib.PgError) as ei:
        z.eval(ctx, np.array([[1.0, 0.0, 2.0]]))
    assert ei.value.code == -5
    # quirks survive the trip to the device
    for s, want1 in (("-5", -5.0), ("2*1e-5", 0.0), ("2^3^2", 64.0), ("7%3", 1.0), ("0#4*2", 8.0)):
        q = pa.Expr(s)
        assert q.eval(ctx, np.zeros((0, 3)))[0] == want1
    # `^` with an integer-valued exponent is math.Pow's repeated-squaring loop (math/pow.go), not a libm pow: exact where the
    # products are, Go's own value where they are not; an integer power stays an integer exponent of the next `^`
    for s, want1 in (("4e2^4e0", 25600000000.0), ("4e2^4e0%1000", 0.0), ("10^308", 1.0000000000000006e308), ("10^309", np.inf),
                     ("(0-238.9)^(4e2^4e0-8e3)", np.inf), ("(0-238.9)^(4e2^4e0-7999)", -np.inf), ("(0-2)^3", -8.0),
                     ("1.5^(0-3)", 1.0 / (1.5 * 1.5 * 1.5)), ("7.25^1", 7.25), ("9^0.5", 3.0), ("16^(0-0.5)", 0.25)):
        q = pa.Expr(s)
        got1 = q.eval(ctx, np.zeros((0, 3)))[0]
        assert got1 == want1 and got1 == o.expr_eval(o.expr_parse(s), lambda name: None), s
        q.free()
    e.free()
    z.free()


# ---------------------------------------------------------------------------------------------
# sort
# ---------------------------------------------------------------------------------------------
def test_sort_reference_known_answers_on_device(ctx, golden):
    for c in golden["sort"]:
        got = ctx.sort_scores(np.array(c["scores"]), descending=c["descending"]).tolist()
        assert got == c["expect_order"], c["ref"]


def test_sort_matches_oracle_segmented(ctx):
    rng = np.random.default_rng(7)
    sizes = [5000, 0, 1, 2, 8192, 777, 20000]                # 20000 > LDS capacity → global path
    segs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    s = rng.random(int(segs[-1]))
    s[::97] = s[1::97][: len(s[::97])]                       # ties
    s[5] = np.nan
    s[11] = -0.0
    s[12] = 0.0
    for desc in (True, False):
        got = ctx.sort_scores(s, segs, descending=desc)
        for i in range(len(sizes)):
            a, b = int(segs[i]), int(segs[i + 1])
            assert np.array_equal(got[a:b], o.sort_scores(s[a:b], desc)), (desc, i)


def _sort_corner_scores(rng, segs, equal_segment=None):
    s = rng.random(int(segs[-1]))
    s[::7] = np.round(s[::7], 1)                             # heavy ties
    s[3::501] = np.nan
    s[10::997] = -0.0
    s[11::997] = 0.0
    if equal_segment is not None:
        s[segs[equal_segment]:segs[equal_segment + 1]] = 0.25   # one segment of identical scores
    return s


def _check_sort(ctx, s, segs, sizes, tag):
    for desc in (True, False):
        got = ctx.sort_scores(s, segs, descending=desc)
        for i in range(len(sizes)):
            a, b = int(segs[i]), int(segs[i + 1])
            assert np.array_equal(got[a:b], o.sort_scores(s[a:b], desc)), (tag, desc, i, sizes[i])


def test_sort_register_network_segments_up_to_8192(ctx):
    """All segments <= 8192: the register-resident bitonic network (shuffle / LDS exchanges) — ties by
    index, NaN last, -0 == +0, heavy duplicates, every power-of-two boundary.  (The split sort would take a call of
    this size: switched off here, it has the test below.)"""
    rng = np.random.default_rng(17)
    sizes = [8192, 0, 1, 2, 3, 511, 512, 513, 5000, 4096, 4097, 64, 65, 1000, 8191]
    segs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    s = _sort_corner_scores(rng, segs, 8)
    ctx.set_option("split_sort_max", 0)
    ctx.set_option("rank_sort_max", 0)
    try:
        _check_sort(ctx, s, segs, sizes, "network")
    finally:
        ctx.set_option("split_sort_max", 96)
        ctx.set_option("rank_sort_max", 32)


def test_sort_few_segments_rank_by_counting(ctx):
    """A few short lists (lists x items^2 under the knob rank_sort_work — lifted here — or up to 8 lists beyond 8192 items)
    take the counting kernel (64 items per workgroup, spread over the chip): same order as the network — ties by index,
    NaN last, -0 == +0, an all-equal segment, the chunk boundaries of the four-way split."""
    rng = np.random.default_rng(19)
    ctx.set_option("rank_sort_work", 1e18)
    try:
        for sizes in ([8192, 0, 1, 5000, 513, 63, 4097, 2], [5000], [7], [8191, 8185], [16384, 9000, 8193], [12345]):
            segs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
            s = _sort_corner_scores(rng, segs, 3 if len(sizes) > 3 else None)
            _check_sort(ctx, s, segs, sizes, "counting")
    finally:
        ctx.set_option("rank_sort_work", 7e7)


def test_sort_split_over_the_chip(ctx):
    """Up to 96 lists of 1025 … 8192 items (more work than the counting kernel takes, switched off here so that the single lists
    come this way too): every 512-slot piece sorted by one wave, final positions by binary searches in the
    list's other runs (csrc/split_sort.hpp) — the same total order as the network: ties by input position across and inside
    runs (a list of identical scores, ties that straddle run boundaries), NaN last, -0 == +0, ragged lists beside full ones."""
    rng = np.random.default_rng(23)
    cases = ([5000], [8192], [1025, 0, 1, 511, 512, 513], [8192, 5000, 1536, 1537, 2, 4097, 8191, 1024],
             [5000] * 32, [3000 + 17 * i for i in range(96)], [8192] * 3 + [1] * 50)
    ctx.set_option("rank_sort_max", 0)
    for sizes in cases:
        segs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
        s = _sort_corner_scores(rng, segs, 1 if len(sizes) > 1 else None)
        if len(sizes) == 1:
            s[500:530] = 0.5                                   # equal scores across the first run boundary
            s[1000:2100] = np.nan                              # NaN items across two boundaries
        _check_sort(ctx, s, segs, sizes, "split")
    before = ctx.stats().sort_split_calls
    ctx.sort_scores(rng.random(5000), descending=True)
    assert ctx.stats().sort_split_calls == before + 1
    ctx.set_option("rank_sort_max", 32)
    ctx.sort_scores(rng.random(5000), descending=True)          # one list of 5 000: counting is the shorter launch
    assert ctx.stats().sort_split_calls == before + 1
    ctx.sort_scores(rng.random(8 * 5000), np.arange(9, dtype=np.uint32) * 5000, descending=True)
    assert ctx.stats().sort_split_calls == before + 2


def test_sort_random_calls_take_every_path(ctx):
    """Random calls — 1 … 200 lists of random sizes up to 9 000, tie densities from none to all-equal, NaN / signed zeros / infinities
    sprinkled in, both directions: whatever path the call's size selects (counting, split, one-workgroup network, the global-memory
    network beyond 8 192 items), the order is the oracle's."""
    rng = np.random.default_rng(31)
    for case in range(24):
        nseg = int(rng.choice([1, 2, 3, 7, 20, 60, 97, 200]))
        top = int(rng.choice([40, 600, 1500, 5000, 8192, 9000]))
        sizes = rng.integers(0, top + 1, nseg).tolist()
        sizes[int(rng.integers(0, nseg))] = top
        segs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
        n = int(segs[-1])
        levels = int(rng.choice([1, 3, 50, 10**9]))
        s = rng.random(n) if levels == 10**9 else rng.integers(0, levels, n).astype(np.float64) / levels
        s[rng.random(n) < 0.002] = np.nan
        s[rng.random(n) < 0.002] = -0.0
        s[rng.random(n) < 0.002] = np.inf
        s[rng.random(n) < 0.002] = -np.inf
        _check_sort(ctx, s, segs, sizes, "random %d" % case)


def test_recall_final_order_split_equals_network_and_oracle(ctx):
    """The top-K's final order (descending score, then row) through the split sort (up to ~450 K items in up to 96 lists, K > 1024) equals the
    one-workgroup network's (the counting kernel's where the call is small) and the oracle's; a table with fewer rows than K fills the tail (~0 rows, -inf scores, count)."""
    rng = np.random.default_rng(29)
    for rows, k in ((120_000, 5000), (3000, 5000), (9000, 8192), (50_000, 1025)):
        tab = rng.uniform(-1, 1, (rows, 128)).astype(np.float32)
        h = (rows // 2) // 3 * 3
        tab[0:h:3] = tab[1:h:3]                                    # equal rows = equal scores: ties by row
        t = pa.Table(ctx, rows, 128)
        t.upload(tab)
        for nq in (1, 8, 33, 54):
            q = rng.uniform(-1, 1, (nq, 128)).astype(np.float32)
            rd, sd, cd = t.recall_topk(q, k)                       # defaults: counting for a small call, else the split sort
            before = ctx.stats().sort_split_calls
            ctx.set_option("rank_sort_max", 0)
            try:
                r1, s1, c1 = t.recall_topk(q, k)
                assert ctx.stats().sort_split_calls > before
                ctx.set_option("split_sort_max", 0)
                r0, s0, c0 = t.recall_topk(q, k)
            finally:
                ctx.set_option("split_sort_max", 96)
                ctx.set_option("rank_sort_max", 32)
            assert np.array_equal(rd, r1) and np.array_equal(sd.view(np.uint32), s1.view(np.uint32)) and np.array_equal(cd, c1)
            assert np.array_equal(r0, r1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32)) and np.array_equal(c0, c1)
            for i in (0, nq - 1):
                er, es = o.recall_topk(tab, q[i : i + 1], k)
                er, es = er[0], es[0]
                n = min(k, rows)
                assert c1[i] == n
                assert np.array_equal(r1[i, :n], er[:n]) and np.array_equal(s1[i, :n].view(np.uint32), es[:n].view(np.uint32))
                assert np.all(r1[i, n:] == np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(np.isneginf(s1[i, n:]))
        del t


# ---------------------------------------------------------------------------------------------
# DPP
# ---------------------------------------------------------------------------------------------
def test_dpp_matches_oracle(ctx):
    rng = np.random.default_rng(8)
    n_tab, d, n = 4000, 128, 500
    centers = rng.standard_normal((12, d)).astype(np.float32)
    tab = (centers[rng.integers(0, 12, n_tab)] + 0.2 * rng.standard_normal((n_tab, d))).astype(np.float32)
    t = pa.Table(ctx, n_tab, d)
    t.upload(tab)
    cand = rng.choice(n_tab, n, replace=False).astype(np.uint32)
    rel = np.sort(rng.random(n))[::-1].copy()
    emb = o.l2_normalize_f64(tab[cand].astype(np.float64))
    L = o.dpp_kernel_matrix(emb, rel, 1.0)
    for topn, window in ((100, 10), (10, 10), (37, 5), (500, 10)):
        want = o.dpp_with_window(L, topn, window)
        got = pa.dpp(ctx, t, cand, rel, 1.0, topn, window, True)
        assert np.array_equal(got, want), (topn, window)
    assert not np.array_equal(pa.dpp(ctx, t, cand, rel, 1.0, 100, 10, True), np.arange(100))
    # the other greedy kernels: 512 < n <= 1024 (sixteen elements per lane), and beyond it / wider windows (one workgroup)
    for n2, topn, window in ((800, 60, 10), (1024, 30, 7), (1100, 40, 10), (600, 40, 20)):
        cand2 = rng.choice(n_tab, n2, replace=False).astype(np.uint32)
        rel2 = np.sort(rng.random(n2))[::-1].copy()
        L2 = o.dpp_kernel_matrix(o.l2_normalize_f64(tab[cand2].astype(np.float64)), rel2, 1.0)
        assert np.array_equal(pa.dpp(ctx, t, cand2, rel2, 1.0, topn, window, True), o.dpp_with_window(L2, topn, window)), (n2, topn, window)
    t.destroy()


def test_dpp_kernel_matrix_bits_on_both_pipes(ctx):
    """DPPSort.KernelMatrix alone (pg_dpp_kernel_matrix_dev): the fp64 MATRIX pipe (v_mfma_f64_16x16x4_f64, the default since round
    5) and the vector-pipe kernel (knob dpp_valu) give the same bits of L = diag(r) F F^T diag(r), and with alpha = 0 (r = 1
    exactly on host and device; exp() is the one operation whose last bit the two may round differently) those are the bits of the
    oracle's k-ascending fma chains — over sizes around the 64-row tiles, widths that are and are not 16 m + 1, rows whose
    exponents spread over 2^±12 (where any other summation order shows), a zero row and signed zeros."""
    from pairec_amd import _lib
    rng = np.random.default_rng(21)
    cases = [(3, 500, 128, True), (2, 64, 128, True), (2, 65, 64, True), (1, 1, 128, True), (2, 129, 16, False), (1, 200, 20, False),
             (2, 63, 3, True), (1, 333, 130, False), (1, 512, 128, True), (1, 100, 15, False), (1, 70, 33, True),
             (9, 70, 16, True), (17, 130, 8, False), (8, 65, 4, True)]        # full rounds of eight requests + a partial one (the XCD placement)
    for R, n, d, norm in cases:
        emb = rng.standard_normal((R, n, d)).astype(np.float32)
        emb *= np.exp2(rng.integers(-12, 13, (R, n, d))).astype(np.float32)          # products of very different magnitudes
        if n > 4:
            emb[0, 2, : d // 2] = -0.0
            if not norm:
                emb[0, 3] = 0.0                                                      # an all-zero row: S = 0, the chain never leaves +0
        rel = np.sort(rng.random((R, n)), axis=1)[:, ::-1].copy()
        d_e, d_r, d_L = ctx.to_device(emb), ctx.to_device(rel), ctx.malloc(R * n * n * 8)
        for alpha in (0.0, 0.7):
            got = {}
            for valu in (0, 1):
                ctx.set_option("dpp_valu", valu)
                L = np.zeros((R, n, n))
                _lib.check(ctx.L.pg_dpp_kernel_matrix_dev(ctx.h, d_e, d_r, R, n, d, alpha, int(norm), d_L))
                ctx.d2h(L, d_L)
                got[valu] = L
            ctx.set_option("dpp_valu", 0)
            assert np.array_equal(got[0].view(np.uint64), got[1].view(np.uint64)), (R, n, d, norm, alpha)
            want = np.stack([o.dpp_kernel_matrix_f(o.dpp_features(emb[q], None, norm, True), rel[q], alpha) for q in range(R)])
            if alpha == 0.0:
                assert np.array_equal(got[0].view(np.uint64), want.view(np.uint64)), (R, n, d, norm, float(np.max(np.abs(got[0] - want))))
            else:
                assert np.allclose(got[0], want, rtol=1e-15, atol=0.0), (R, n, d, norm)
        for p in (d_e, d_r, d_L):
            ctx.free(p)


def test_dpp_options_match_oracle(ctx):
    """DPPSort.KernelMatrix's switches (dpp_sort.go:382-447): dpp_norm_relevance_score 1 / 2, hook embeddings
    prepended to the table embedding, hook-only rows with and without EnsurePositiveSim, un-normalised rows —
    pick sequences and the relevance scores reported as "dpp_relevance_score" equal the oracle's."""
    rng = np.random.default_rng(18)
    n_tab, d, n, h = 3000, 64, 300, 48       # (a hook-only kernel has rank h + 1: keep topn below it, beyond it the
                                              #  greedy picks by rounding noise in the reference as well)
    centers = rng.standard_normal((10, d)).astype(np.float32)
    tab = (centers[rng.integers(0, 10, n_tab)] + 0.25 * rng.standard_normal((n_tab, d))).astype(np.float32)
    t = pa.Table(ctx, n_tab, d)
    t.upload(tab)
    cand = rng.choice(n_tab, n, replace=False).astype(np.uint32)
    rel = np.sort(rng.random(n))[::-1].copy()
    hook = rng.standard_normal((n, h))
    for has_table, hk, norm, pos, mode, topn, window, alpha in [
            (True, None, True, True, 1, 50, 10, 1.0),
            (True, None, True, True, 2, 50, 10, 2.0),
            (True, hook, True, True, 0, 40, 7, 1.0),
            (False, hook, True, True, 0, 40, 10, 1.0),
            (False, hook, False, False, 2, 30, 5, 0.05),
            (True, None, False, True, 0, 30, 10, 0.1)]:
        rs, ok = o.dpp_relevance(rel, mode)
        assert ok
        F = o.dpp_features(tab[cand] if has_table else None, hk, norm, pos)
        want = o.dpp_with_window(o.dpp_kernel_matrix_f(F, rs, alpha), topn, window)
        got, used = pa.dpp_ex(ctx, t if has_table else None, cand, rel, alpha, topn, window, norm, pos, mode, hk)
        assert np.array_equal(got, want), (has_table, hk is not None, norm, pos, mode)
        assert np.array_equal(used.view(np.uint64), rs.view(np.uint64))
    # "all item score is zero": the reference returns the items unchanged; here the call says so
    with pytest.raises(pa._lib.PgError) as ei:
        pa.dpp_ex(ctx, t, cand, np.full(n, 0.25), 1.0, 10, 10, norm_relevance_score=1)
    assert ei.value.code == -5
    t.destroy()


# ---------------------------------------------------------------------------------------------
# SSD
# ---------------------------------------------------------------------------------------------
def test_ssd_matches_oracle(ctx):
    """SSDSort.SSDWithSlidingWindow (ssd_sort.go:346-486): the pick sequence equals the oracle's for
    every window / gamma / normalisation mode; bar = exact indices (fp64, same operation order)."""
    rng = np.random.default_rng(9)
    n_tab, d, n = 4000, 128, 500
    centers = rng.standard_normal((12, d)).astype(np.float32)
    tab = (centers[rng.integers(0, 12, n_tab)] + 0.2 * rng.standard_normal((n_tab, d))).astype(np.float32)
    t = pa.Table(ctx, n_tab, d)
    t.upload(tab)
    cand = rng.choice(n_tab, n, replace=False).astype(np.uint32)
    rel = np.sort(rng.random(n))[::-1].copy()
    for topn, window, gamma, norm_emb, pos, mode, star in [
            (100, 5, 0.25, True, True, 0, False),      # SSDSortConfig defaults (recconf.go:980-1000)
            (100, 10, 0.5, True, False, 1, False),     # z-scored quality
            (37, 3, 0.25, True, True, 2, False),       # min-max quality
            (60, 5, 0.25, False, True, 0, True),       # SSD*, raw embeddings
            (500, 5, 0.25, True, True, 0, False),      # every candidate picked
            (10, 1, 0.25, True, True, 0, False)]:      # window <= 1 → 5
        emb = o.ssd_embeddings(tab[cand], norm_emb, pos)
        qual, ok = o.ssd_quality(rel, mode)
        assert ok
        want = o.ssd_window(emb, qual, gamma, topn, window, star)
        got, gq = pa.ssd(ctx, t, cand, rel, gamma, topn, window, norm_emb, pos, mode, star)
        assert np.array_equal(got, want), (topn, window, gamma, mode, star)
        assert np.array_equal(gq, qual)
    got, _ = pa.ssd(ctx, t, cand, rel, 0.25, 100, 5)
    assert not np.array_equal(got, np.arange(100))                # diversity changed the order
    # "all item score are zeros": the reference returns the items unchanged
    got, _ = pa.ssd(ctx, t, cand, np.zeros(n), 0.25, 100, 5, norm_quality_score=1)
    assert np.array_equal(got, np.arange(n))
    # a single candidate, and topn larger than n
    got, _ = pa.ssd(ctx, t, cand[:1], rel[:1], 0.25, 10, 5)
    assert got.tolist() == [0]
    t.destroy()


# ---------------------------------------------------------------------------------------------
# typed feature columns on the device
# ---------------------------------------------------------------------------------------------
def test_expression_over_feature_columns_matches_oracle(ctx):
    """pg_features_eval_dev: an expression whose variables are feature columns, per candidate row in fp64 — the numeric
    `expression` normalizer of a new_feature over item features (service/feature/new_feature_op.go:54-115) for a batch — against
    the oracle's ExprASTResult (default grammar) and antlr_result (the antlr subset), incl. rows outside the store (column
    defaults), every column type, a constant expression, an unknown column and a division by zero."""
    rng = np.random.default_rng(33)
    n = 4000
    fs = pa.Features(ctx, n)
    cols = {"clicks": (pa.F_I32, rng.integers(0, 10_000, n).astype(np.int32), 0.0),
            "shows": (pa.F_I64, rng.integers(1, 2**40, n).astype(np.int64), 1.0),
            "price": (pa.F_F32, (rng.random(n) * 100).astype(np.float32), 9.5),
            "ctr": (pa.F_F64, rng.random(n), 0.25)}
    for name, (dt, v, d) in cols.items():
        fs.set_column(name, dt, v, default=d)
    rows = np.concatenate([rng.integers(0, n, 500), [0xFFFFFFFF, n, n - 1, 0]]).astype(np.uint32)

    def column(name, i):
        dt, v, d = cols[name]
        return float(v[rows[i]]) if rows[i] < n else d
    for src in ["(${clicks} + 1) / (${shows} + 2) * ${price}^0.5", "(${clicks} + 1) / (${shows} + 2) * ${price}", "${ctr} * 100 - ${price} # 3",
                "${price} % 7 + ${clicks}", "2.5", "${ctr}"]:
        e = pa.Expr(src)
        got = fs.eval_expr(e, rows)
        ast = o.expr_parse(src)
        want = np.array([o.expr_eval(ast, lambda nm, i=i: column(nm, i)) for i in range(len(rows))])
        if "^" in src:
            assert np.allclose(got, want, rtol=4e-16, atol=0.0), src          # pow: the device's last ulp
        else:
            assert np.array_equal(got, want), src
        e.free()
    src = "(${clicks} + 2*${ctr}) * ${price}^0.1 - ${shows} / 3"
    e = pa.Expr(src, "antlr")
    got = fs.eval_expr(e, rows)
    tree = o.antlr_parse(src)
    want = np.array([o.antlr_result(tree, {nm: column(nm, i) for nm in cols}) for i in range(len(rows))])
    assert np.allclose(got, want, rtol=1e-15, atol=0.0), "antlr"
    e.free()
    e = pa.Expr("${clicks} + ${nope}")
    with pytest.raises(RuntimeError, match="nope.*not a column"):
        fs.eval_expr(e, rows)
    e.free()
    e = pa.Expr("${price} / (${clicks} - ${clicks})")
    with pytest.raises(RuntimeError):
        fs.eval_expr(e, rows)
    e.free()
    fs.destroy()


def test_feature_columns_gather_and_defaults(ctx):
    """pg_features_*: typed columns keyed by item row; an item without the feature (row past the store)
    reads the column default — the device form of feature.defaultValue (algo_data.go:154-171).  Integer
    gathers are exact; the float path is fmaf(value, scale, bias) in fp32."""
    rng = np.random.default_rng(21)
    n = 5000
    fs = pa.Features(ctx, n)
    cat = rng.integers(0, 1000, n).astype(np.int32)
    big = rng.integers(-2**40, 2**40, n).astype(np.int64)
    price = rng.random(n).astype(np.float32) * 100
    ctr = rng.random(n)
    fs.set_column("cat", pa.F_I32, cat, default=0)
    fs.set_column("big", pa.F_I64, big, default=-7)
    fs.set_column("price", pa.F_F32, price, default=0.0)
    fs.set_column("ctr", pa.F_F64, ctr, default=0.5)
    fs.set_column("empty", pa.F_I32, None, default=3)              # declared, no values yet
    assert fs.index("price") == 2 and fs.index("nope") == -1
    rows = np.concatenate([rng.integers(0, n, 300), [0xFFFFFFFF, n, n - 1, 0]]).astype(np.uint32)
    inside = rows < n
    gi = fs.gather_i32(["cat", "big", "empty"], rows)
    want_cat = np.where(inside, cat[np.minimum(rows, n - 1)], 0)
    want_big = np.clip(np.where(inside, big[np.minimum(rows, n - 1)], -7), -2**31, 2**31 - 1)
    assert np.array_equal(gi[:, 0], want_cat) and np.array_equal(gi[:, 1], want_big) and np.all(gi[:, 2] == 3)
    scale = np.array([0.01, 2.0, 1.0], np.float32)
    bias = np.array([0.0, -1.0, 0.25], np.float32)
    gf = fs.gather_f32(["price", "ctr", "cat"], rows, scale, bias)
    cols = [np.where(inside, price[np.minimum(rows, n - 1)], np.float32(0.0)).astype(np.float32),
            np.where(inside, ctr[np.minimum(rows, n - 1)], 0.5).astype(np.float32),
            want_cat.astype(np.float32)]
    for f in range(3):
        want = (cols[f].astype(np.float64) * float(scale[f]) + float(bias[f])).astype(np.float32)   # fmaf in fp32
        assert np.array_equal(gf[:, f], want), f
    assert np.array_equal(fs.gather_f32(["price"], rows)[:, 0], cols[0])                       # no normalizer
    # replacing a column with another dtype keeps its index
    fs.set_column("cat", pa.F_I64, cat.astype(np.int64), default=0)
    assert fs.index("cat") == 0 and np.array_equal(fs.gather_i32(["cat"], rows)[:, 0], want_cat)
    with pytest.raises(RuntimeError):
        fs.gather_i32(["price"], rows)                              # not an integer column
    fs.destroy()


def test_rank_fm2t_from_candidate_rows(ctx):
    """pg_rank_fm2t_rows_dev: the item field ids are assembled on the device from 8 integer feature
    columns; scores are bit-identical to handing the same ids over pre-resolved."""
    fw = o.Fm2tWeights(vocab=3000)
    m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, pa.PREC_F32, pa.pack_fm2t(fw))
    rng = np.random.default_rng(5)
    n_items_tab = 20000
    fs = pa.Features(ctx, n_items_tab)
    fields = rng.integers(0, 3000, (n_items_tab, 8)).astype(np.int32)
    names = ["item_field_%d" % f for f in range(8)]
    for f in range(8):
        fs.set_column(names[f], pa.F_I32 if f % 2 == 0 else pa.F_I64, fields[:, f], default=0)
    sizes = [3000, 129, 1]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    users = o.synth_rows(o.SEED_QUERY, 9, 3, 128)
    ufids = rng.integers(0, 3000, (3, 8)).astype(np.int32)
    cand = rng.integers(0, n_items_tab, int(off[-1])).astype(np.uint32)
    cand[7] = 0xFFFFFFFF                                            # an item without features: defaults (id 0)
    ifids = np.where((cand < n_items_tab)[:, None], fields[np.minimum(cand, n_items_tab - 1)], 0).astype(np.int32)
    got = m.rank_fm2t_rows(fs, names, users, ufids, cand, off)
    ref = m.rank_fm2t(users, ufids, ifids, off)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    want = np.concatenate([o.fm2t_forward(fw, 0, users[r], ufids[r], ifids[off[r]:off[r + 1]]) for r in range(3)])
    assert np.max(np.abs(got.astype(np.float64) - want)) <= 3e-7
    fs.destroy()
    m.destroy()


# ---------------------------------------------------------------------------------------------
# concurrency: pairec calls its plugins from many goroutines (rank_service.go:163-166 fans out 50)
# ---------------------------------------------------------------------------------------------
def test_concurrent_callers_share_one_context(ctx):
    """Several host threads drive the same context at once (ctypes drops the GIL during the calls): the
    context serialises them on its stream and every caller gets its own, correct answer."""
    import threading
    n, d, k = 120000, 128, 300
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    w = o.Dnn3Weights()
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    errors = []

    def worker(i):
        try:
            q = o.synth_rows(o.SEED_QUERY, 10 * i, 3 + i, d)
            for _ in range(3):
                rows, scores, _ = t.recall_topk(q, k)
                orow, osc = o.recall_topk(tab, q, k)
                assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
                cand = rows[0].astype(np.uint32)
                got = m.rank_dnn3(t, q[:1], cand, [0, k])
                want = o.dnn3_forward(w, 0, q[0], tab[cand.astype(np.int64)])
                assert np.max(np.abs(got.astype(np.float64) - want)) <= 2e-7
                order = ctx.sort_scores(got.astype(np.float64))
                assert np.array_equal(order, o.sort_scores(got.astype(np.float64), True))
        except Exception as e:                                 # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    m.destroy()
    t.destroy()


# ---------------------------------------------------------------------------------------------
# BASELINE.json's full size: 100 M x 128 table, K = 5 000, 256 requests — size-independent properties
# ---------------------------------------------------------------------------------------------
def test_full_size_pipeline_properties(ctx):
    """The benchmark configuration itself.  The oracle cannot scan 100 M rows x 256 queries in seconds, so:
    (a) order: scores non-increasing, ties by row, rows unique and in range;
    (b) exactness: returned scores equal the oracle's dot products of the returned rows, bit for bit;
    (c) completeness on samples: in random 1 M-row slices (regenerated on the CPU from the synthetic
        generator), every row whose key beats the K-th returned key is in the answer;
    (d) batching invariance: a query alone / in a batch of 128 / in the batch of 256 gets the same answer;
    (e) rank + fusion + sort over all 1.28 M candidates: a sample of scores within tolerance of the
        oracle, the per-request order equal to the oracle's sort of the device's own scores;
    (f) the plan `bench.py` times: once the table's threshold model has observed 1 024 verified queries it predicts the
        first thresholds (plan 0) — four more batches train it, then the SAME 256 queries on predicted thresholds return
        the pilot plan's answer bit for bit, and a fresh batch passes the exactness and completeness checks."""
    n, d, k, R = 100_000_000, 128, 5000, 256
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    q = o.synth_rows(o.SEED_QUERY, 0, R, d)
    rows, scores, cnt = t.recall_topk(q, k)
    assert cnt.tolist() == [k] * R
    # (a)
    assert rows.max() < n
    ds = np.diff(scores.astype(np.float64), axis=1)
    assert np.all(ds <= 0)
    tie = ds == 0
    assert np.all(np.diff(rows.astype(np.int64), axis=1)[tie] > 0)
    for r in range(0, R, 37):
        assert len(np.unique(rows[r])) == k
    # (b) on 4 queries
    for r in (0, 101, 255, 128):
        got_rows = t.gather(rows[r].astype(np.uint32))
        for j in (0, 1, 2500, 4999):                            # the gather itself, against the generator
            rr = int(rows[r, j])
            assert np.array_equal(got_rows[j], o.synth_rows(o.SEED_TABLE, rr, 1, d)[0])
        want = o.dot_scores(got_rows, q[r:r + 1])[0]
        assert np.array_equal(bits(scores[r]), bits(want))
    # (c) three random slices, 6 queries each
    rng = np.random.default_rng(77)
    for _ in range(3):
        lo = int(rng.integers(0, n - 1_000_000))
        sl = o.synth_rows(o.SEED_TABLE, lo, 1_000_000, d)
        qs = [0, 3, 64, 127, 200, 255]
        sc = o.dot_scores(sl, q[qs])
        for j, r in enumerate(qs):
            kth_s, kth_row = scores[r, -1], int(rows[r, -1])
            s = sc[j]
            better = np.nonzero((s > kth_s) | ((s == kth_s) & (np.arange(lo, lo + len(s)) < kth_row)))[0] + lo
            assert np.all(np.isin(better, rows[r])), (lo, r)
            inside = rows[r][(rows[r] >= lo) & (rows[r] < lo + len(s))]
            assert len(inside) == len(better) or (len(inside) == len(better) + 1 and kth_row in inside)
    # (d)
    r1, s1, _ = t.recall_topk(q[5:6], k)
    assert np.array_equal(r1[0], rows[5]) and np.array_equal(bits(s1[0]), bits(scores[5]))
    r128, s128, _ = t.recall_topk(q[:128], k)
    assert np.array_equal(r128, rows[:128]) and np.array_equal(bits(s128), bits(scores[:128]))
    # (e)
    w = o.Dnn3Weights()
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    off = (np.arange(R + 1) * k).astype(np.uint32)
    dnn = m.rank_dnn3(t, q, rows.reshape(-1).astype(np.uint32), off)
    for r in (0, 77, 255):
        idx = rng.integers(0, k, 40)
        cand = rows[r, idx].astype(np.uint32)
        want = o.dnn3_forward(w, 1, q[r], t.gather(cand))
        assert np.max(np.abs(dnn[r * k + idx].astype(np.float64) - want)) <= 1e-5
    ex = pa.Expr("${gpu_dnn}*(1+${current_score})^0.1")
    cols = {"gpu_dnn": dnn.astype(np.float64), "current_score": scores.reshape(-1).astype(np.float64)}
    fused = ex.eval(ctx, np.stack([cols[v] for v in ex.var_names]))
    order = ctx.sort_scores(fused, off, descending=True)
    for r in (0, 13, 254):
        seg = fused[r * k:(r + 1) * k]
        assert np.array_equal(order[r * k:(r + 1) * k], o.sort_scores(seg, True))
    fs = fused.reshape(R, k)
    sorted_scores = np.take_along_axis(fs, order.reshape(R, k).astype(np.int64), axis=1)
    assert np.all(np.diff(sorted_scores, axis=1) <= 0)
    m.destroy()
    # (f) plan 0 at full size
    p0 = ctx.stats().recall_predicted
    for b in range(5):
        t.recall_topk(o.synth_rows(o.SEED_QUERY, 1000 + 256 * b, R, d), k)      # other users: 1 280 more verified queries
    rows2, scores2, cnt2 = t.recall_topk(q, k)
    assert ctx.stats().recall_predicted > p0, "the threshold model never predicted: plan 0 was not exercised"
    assert np.array_equal(rows2, rows) and np.array_equal(bits(scores2), bits(scores)) and cnt2.tolist() == [k] * R
    qn = o.synth_rows(o.SEED_QUERY, 5000, R, d)
    p1 = ctx.stats().recall_predicted
    rows3, scores3, cnt3 = t.recall_topk(qn, k)
    assert ctx.stats().recall_predicted > p1 and cnt3.tolist() == [k] * R
    for r in (0, 200):
        want = o.dot_scores(t.gather(rows3[r].astype(np.uint32)), qn[r:r + 1])[0]
        assert np.array_equal(bits(scores3[r]), bits(want))
    lo = int(rng.integers(0, n - 1_000_000))
    sl = o.synth_rows(o.SEED_TABLE, lo, 1_000_000, d)
    qs = [1, 100, 255]
    sc = o.dot_scores(sl, qn[qs])
    for j, r in enumerate(qs):
        kth_s, kth_row = scores3[r, -1], int(rows3[r, -1])
        s = sc[j]
        better = np.nonzero((s > kth_s) | ((s == kth_s) & (np.arange(lo, lo + len(s)) < kth_row)))[0] + lo
        assert np.all(np.isin(better, rows3[r])), (lo, r)
    t.destroy()


# ---------------------------------------------------------------------------------------------
# the whole hot path in one call
# ---------------------------------------------------------------------------------------------
def test_recommend_one_call_equals_the_stages(ctx):
    """pg_recommend_dnn3_dev = recall → rank → fusion → sort behind one ABI call: every output equals what
    the stage-by-stage calls produce, and the page order equals the oracle's sort of the fused scores."""
    n, d, k, R = 150000, 128, 400, 37
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    w = o.Dnn3Weights()
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    ex = pa.Expr("${gpu_dnn}*(1+${current_score})^0.1")
    q = o.synth_rows(o.SEED_QUERY, 11, R, d)
    N = R * k
    d_q = ctx.to_device(q)
    d_rows, d_sc, d_rk, d_fu, d_or = (ctx.malloc(N * 8), ctx.malloc(N * 4), ctx.malloc(N * 4), ctx.malloc(N * 8),
                                      ctx.malloc(N * 4))
    pa._lib.check(ctx.L.pg_recommend_dnn3_dev(ctx.h, t.h, m.h, ex.h, b"gpu_dnn", d_q, R, k, d_rows, d_sc, d_rk, d_fu, d_or, None))
    rows, sc, rk = np.zeros((R, k), np.uint64), np.zeros((R, k), np.float32), np.zeros((R, k), np.float32)
    fu, order = np.zeros((R, k), np.float64), np.zeros((R, k), np.uint32)
    for a, p in ((rows, d_rows), (sc, d_sc), (rk, d_rk), (fu, d_fu), (order, d_or)):
        ctx.d2h(a, p)
    # stage by stage
    r2, s2, _ = t.recall_topk(q, k)
    assert np.array_equal(rows, r2) and np.array_equal(bits(sc), bits(s2))
    off = (np.arange(R + 1) * k).astype(np.uint32)
    rk2 = m.rank_dnn3(t, q, r2.reshape(-1).astype(np.uint32), off)
    assert np.array_equal(bits(rk.reshape(-1)), bits(rk2))
    cols = {"gpu_dnn": rk2.astype(np.float64), "current_score": s2.reshape(-1).astype(np.float64)}
    fu2 = ex.eval(ctx, np.stack([cols[v] for v in ex.var_names]))
    assert np.array_equal(fu.reshape(-1).view(np.uint64), fu2.view(np.uint64))
    for r in range(R):
        assert np.array_equal(order[r], o.sort_scores(fu[r], True))
    # a RankScore that names an unknown variable is refused
    bad = pa.Expr("${gpu_dnn}+${ctr}")
    with pytest.raises(RuntimeError):
        pa._lib.check(ctx.L.pg_recommend_dnn3_dev(ctx.h, t.h, m.h, bad.h, b"gpu_dnn", d_q, R, k, d_rows, d_sc, d_rk, d_fu, d_or, None))
    for p in (d_q, d_rows, d_sc, d_rk, d_fu, d_or):
        ctx.free(p)
    m.destroy()
    t.destroy()


def test_recall_threshold_model_predicts_then_backs_off_on_queries_it_does_not_describe(ctx):
    """After 1024 observed queries of one K the screening threshold comes from the table's threshold model instead of
    the pilot sample (csrc/recall.hip): mean + z * sd of the query's score distribution, z learnt from
    the K-th best scores of verified batches.  Exactness never depends on it — a threshold that leaves fewer than K
    candidates fails the same verification as a sampled one.  Forced on for a small table: (1) it takes over and the
    answers stay the oracle's; (2) queries along a two-valued column, whose K-th best score sits far below what the
    Gaussian-tail model says, are caught, re-run and answered exactly; (3) the table then stays on the pilot plan."""
    rng = np.random.default_rng(47)
    n, d, k, nq = 600_000, 128, 300, 160
    tab = rng.standard_normal((n, d)).astype(np.float32)
    tab[:, 127] = np.where(rng.random(n) < 0.5, np.float32(3.0), np.float32(-3.0))
    for name, v in (("predict_min_rows", "0"), ("pilot_fraction", "0.125")):
        ctx.set_option(name, v)
    try:
        t = pa.Table(ctx, n, d)
        t.upload(tab)
        s0 = ctx.stats()
        batches = []
        for b in range(10):
            q = rng.standard_normal((nq, d)).astype(np.float32)
            q[:, 127] = 0.0
            rows, scores, _ = t.recall_topk(q, k)
            batches.append((q, rows, scores))
        s1 = ctx.stats()
        assert s1.recall_rescans == s0.recall_rescans
        assert s1.recall_predicted - s0.recall_predicted >= 2, "the model never took over"
        for q, rows, scores in (batches[0], batches[-1]):
            orow, osc = o.recall_topk(tab, q, k)
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
        # (1b) a lone request rides the same model (its full pass streams the 4-bit shadow when the table has one)
        for name, v in (("i4_min_rows", "0"),):
            ctx.set_option(name, v)
        q1 = rng.standard_normal((2, d)).astype(np.float32)
        q1[:, 127] = 0.0
        sa = ctx.stats()
        rows, scores, _ = t.recall_topk(q1[:1], k)
        rows2, scores2, _ = t.recall_topk(q1, k)
        sb = ctx.stats()
        ctx.set_option("i4_min_rows", str(1 << 22))
        assert sb.recall_predicted - sa.recall_predicted == 2 and sb.recall_rescans == sa.recall_rescans
        orow, osc = o.recall_topk(tab, q1, k)
        assert np.array_equal(rows, orow[:1]) and np.array_equal(bits(scores), bits(osc[:1]))
        assert np.array_equal(rows2, orow) and np.array_equal(bits(scores2), bits(osc))
        s1 = ctx.stats()
        # (2) three queries of the batch look along the two-valued column: half the table scores +3|q|, the K-th best
        # is ~1 sd above the mean where the model expects ~3.3
        q = rng.standard_normal((nq, d)).astype(np.float32)
        q[:, 127] = 0.0
        for i in (5, 77, 159):
            q[i] = 0.0
            q[i, 127] = 1.0 + i
            q[i, :8] = 0.01 * rng.standard_normal(8)           # (break the ties between the 300 K rows at +3)
        rows, scores, _ = t.recall_topk(q, k)
        s2 = ctx.stats()
        assert s2.recall_rescans == s1.recall_rescans + 1 and s2.recall_predicted == s1.recall_predicted
        orow, osc = o.recall_topk(tab, q, k)
        assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
        # (3) backed off: the next batches run the pilot plan (and teach the model again)
        q, rows0, scores0 = batches[3]
        rows, scores, _ = t.recall_topk(q, k)
        s3 = ctx.stats()
        assert s3.recall_predicted == s2.recall_predicted and s3.recall_rescans == s2.recall_rescans
        assert np.array_equal(rows, rows0) and np.array_equal(bits(scores), bits(scores0))
        # the knob turns it off
        ctx.set_option("no_predict", "1")
        rows, scores, _ = t.recall_topk(q, k)
        ctx.set_option("no_predict", "0")
        assert np.array_equal(rows, rows0) and np.array_equal(bits(scores), bits(scores0))
        t.destroy()
    finally:
        ctx.set_option("predict_min_rows", str(1 << 22))
        ctx.set_option("pilot_fraction", "0")
