"""GPU tests of the request coalescer (pg_coalescer_*): single-request calls from many host threads must give
bit-identical answers to the same requests issued alone, and share table passes (SURVEY.md 8b "Threading";
service/recall.go:129-145, service/rank/rank_service.go:264-289 are the concurrent call sites it serves)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import pairec_amd as pa
from oracle import oracle as o

pytestmark = pytest.mark.gpu

EXPR = "${gpu_dnn}*(1+${current_score})^0.1"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def run_threads(n, fn):
    """fn(i) on n threads, released together by a barrier (Python starts threads one by one, far slower than the
    coalescer's window); re-raises the first exception."""
    errs = []
    gate = threading.Barrier(n)

    def wrap(i):
        try:
            gate.wait()
            fn(i)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=wrap, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]


@pytest.fixture(scope="module")
def world(ctx):
    n, d = 400_000, 128
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    w = o.Dnn3Weights()
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    ex = pa.Expr(EXPR)
    yield t, m, ex
    m.destroy()
    t.destroy()


def test_coalesced_recommend_equals_solo_calls(ctx, world):
    """256 threads, one request each (twice over): the page every caller receives equals the first top_n entries of
    the same request through pg_recommend_dnn3_dev alone — ids, order and every score bit for bit."""
    t, m, ex = world
    k, top_n, callers = 500, 100, 256
    q = o.synth_rows(o.SEED_QUERY, 0, 2 * callers, 128)
    # solo reference: one request per call
    ref = []
    for i in range(0, 2 * callers, 37):          # a sample of the requests, each alone
        rows, rec, rnk, fus, order, cnt = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q[i:i + 1], k)
        p = order[0][:top_n]
        ref.append((i, rows[0][p], rec[0][p], rnk[0][p], fus[0][p]))
    co = pa.Coalescer(ctx, t, k, m, ex, "gpu_dnn", max_top_n=top_n, max_wait_us=2000)
    got = [None] * (2 * callers)

    def call(i):
        for j in (i, i + callers):
            got[j] = co.recommend(q[j], top_n)
    run_threads(callers, call)
    st = co.stats()
    co.destroy()
    for i, rows, rec, rnk, fus in ref:
        g = got[i]
        assert g[4] == top_n
        assert np.array_equal(g[0], rows), "request %d: page ids / order differ from the solo call" % i
        assert np.array_equal(bits(g[1]), bits(rec)) and np.array_equal(bits(g[2]), bits(rnk))
        assert np.array_equal(bits(g[3]), bits(fus))
    assert st.requests[2] == 2 * callers
    # the calls shared passes: far fewer batches than requests
    assert st.batches[2] <= 2 * callers // 8, "requests were not coalesced: %d batches" % st.batches[2]
    assert st.largest_batch[2] >= 32


def test_coalesced_recall_and_rank_equal_direct_calls(ctx, world):
    """The recall flavour (VectorRecall → IAlgorithm.Run) and the rank flavour (one 100-item batch per call, as
    RankService.Rank issues them) against pg_recall_topk / pg_rank_dnn3 on the same inputs."""
    t, m, ex = world
    k, callers = 300, 64
    q = o.synth_rows(o.SEED_QUERY, 500, callers, 128)
    rows_ref, sc_ref, _ = t.recall_topk(q, k)
    co = pa.Coalescer(ctx, t, k, m, max_rank_items=100, max_wait_us=1000)
    got = [None] * callers
    run_threads(callers, lambda i: got.__setitem__(i, co.recall(q[i])))
    for i in range(callers):
        assert got[i][2] == k
        assert np.array_equal(got[i][0], rows_ref[i]) and np.array_equal(bits(got[i][1]), bits(sc_ref[i]))
    # rank: every caller scores its request's candidates in three batches of 100 (BatchCount)
    ranks = [None] * callers

    def rank(i):
        cand = rows_ref[i].astype(np.uint32)
        ranks[i] = np.concatenate([co.rank_dnn3(q[i], cand[b:b + 100]) for b in range(0, k, 100)])
    run_threads(callers, rank)
    st = co.stats()
    co.destroy()
    off = (np.arange(callers + 1) * k).astype(np.uint32)
    ref = m.rank_dnn3(t, q, rows_ref.reshape(-1).astype(np.uint32), off).reshape(callers, k)
    for i in range(callers):
        assert np.array_equal(bits(ranks[i]), bits(ref[i])), "rank scores of caller %d differ" % i
    assert st.requests[1] == callers * 3 and st.batches[1] < callers * 3
    # argument checks
    co2 = pa.Coalescer(ctx, t, k, m, max_rank_items=100)
    with pytest.raises(pa._lib.PgError):
        co2.rank_dnn3(q[0], np.arange(101, dtype=np.uint32))           # more than max_rank_items
    with pytest.raises(pa._lib.PgError):
        co2.rank_dnn3(q[0], np.array([t.rows], dtype=np.uint32))        # row outside the table
    with pytest.raises(pa._lib.PgError):
        co2.recommend(q[0], 10)                                         # no RankScore expression
    co2.destroy()


def test_coalescer_survives_a_failed_recall_plan(ctx):
    """A table of identical rows defeats the pilot and the growing-chunk plan (every row ties with every
    threshold): the batch's verification fails after it has run, the completer re-runs it with the next plans, and
    the callers still get the exact answer (lowest rows first)."""
    n, d, k = 2_200_000, 64, 100
    tab = np.zeros((n, d), dtype=np.float32)
    tab[:, 5] = 1.0
    t = pa.Table(ctx, n, d)
    t.upload(tab)
    co = pa.Coalescer(ctx, t, k, max_wait_us=500)
    q = np.zeros((8, d), dtype=np.float32)
    q[:, 5] = 1.0
    got = [None] * 8
    run_threads(8, lambda i: got.__setitem__(i, co.recall(q[i])))
    st = co.stats()
    co.destroy()
    for g in got:
        assert g[0].tolist() == list(range(k)) and np.all(g[1] == 1.0)
    assert st.replans >= 1
    t.destroy()


def test_coalescer_mixed_flavours_errors_and_shutdown(ctx, world):
    """All three flavours in flight at once from different threads; a RankScore that divides by zero fails only the
    requests it concerns (the reference panics in ExprASTResult; here PG_ERR_ARITH for that caller); destroying the
    coalescer while callers are queued fails them cleanly instead of hanging."""
    t, m, ex = world
    k = 200
    q = o.synth_rows(o.SEED_QUERY, 900, 96, 128)
    rows_ref, sc_ref, _ = t.recall_topk(q, k)
    co = pa.Coalescer(ctx, t, k, m, ex, "gpu_dnn", max_top_n=50, max_rank_items=100, max_wait_us=500)
    out = [None] * 96

    def work(i):
        if i % 3 == 0:
            out[i] = ("recall", co.recall(q[i]))
        elif i % 3 == 1:
            out[i] = ("rank", co.rank_dnn3(q[i], rows_ref[i][:100].astype(np.uint32)))
        else:
            out[i] = ("recommend", co.recommend(q[i], 50))
    run_threads(96, work)
    off = np.array([0, 100], dtype=np.uint32)
    for i in range(96):
        kind, r = out[i]
        if kind == "recall":
            assert np.array_equal(r[0], rows_ref[i]) and np.array_equal(bits(r[1]), bits(sc_ref[i]))
        elif kind == "rank":
            assert np.array_equal(bits(r), bits(m.rank_dnn3(t, q[i:i + 1], rows_ref[i][:100].astype(np.uint32), off)))
        else:
            assert r[4] == 50 and set(r[0].tolist()) <= set(rows_ref[i].tolist()) and np.all(np.diff(r[3]) <= 0)
    st = co.stats()
    assert st.requests[0] == 32 and st.requests[1] == 32 and st.requests[2] == 32
    co.destroy()
    # division by zero: 1 / (current_score - s) is undefined for the candidate whose recall score equals s exactly
    s_hit = float(sc_ref[5][3])
    bad = pa.Expr("${gpu_dnn}/(${current_score}-%r)" % s_hit)
    co = pa.Coalescer(ctx, t, k, m, bad, "gpu_dnn", max_top_n=10, max_wait_us=2000)
    res = [None] * 8

    def call(i):
        try:
            res[i] = co.recommend(q[i], 10)
        except pa._lib.PgError as e_:
            res[i] = e_
    run_threads(8, call)
    assert isinstance(res[5], pa._lib.PgError) and res[5].code == -5 and "division by zero" in str(res[5])
    assert all(not isinstance(res[i], Exception) for i in range(8) if i != 5), "one request's arithmetic error leaked into the batch"
    co.destroy()
    bad.free()
    # shutdown with callers queued: a long wait and one slot keep requests in the queue while destroy runs
    co = pa.Coalescer(ctx, t, k, max_wait_us=200_000, depth=1)
    got = []

    def late(i):
        try:
            got.append(co.recall(q[i])[2])
        except pa._lib.PgError as e_:
            got.append(e_)
    th = [threading.Thread(target=late, args=(i,)) for i in range(6)]
    for x in th:
        x.start()
    import time
    time.sleep(0.05)
    co.destroy()
    for x in th:
        x.join(30)
        assert not x.is_alive(), "a caller is still blocked after pg_coalescer_destroy"
    assert len(got) == 6 and all(g == k or isinstance(g, pa._lib.PgError) for g in got)


def test_partial_fallback_reruns_only_the_failed_requests(ctx):
    """With a deliberately thin pilot margin (pilot_sigmas 0.5 instead of 6) the sampled threshold is too high for a
    few requests of a batch: their candidate lists come up short, the verification notices, and only those requests
    are re-run (recall from the growing-chunk plan + rank + fusion + sort) — every answer stays exact, through the
    batch call, the recall call and the coalescer."""
    n, d, k, R = 2_300_000, 64, 100, 64
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    w = o.Dnn3Weights(d_user=64, d_item=64, h1=256, h2=128)
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 64))
    ex = pa.Expr(EXPR)
    ctx.set_option("pilot_sigmas", 0.5)
    ctx.set_option("pilot_fraction", 0.5)          # (half the table as the sample: the +8 in K' then does not cover the thin margin)
    try:
        before = ctx.stats().recall_rescans
        hit = 0
        for b in range(4):
            q = o.synth_rows(o.SEED_QUERY, 64 * b, R, d)
            r0 = ctx.stats().recall_rescans
            rows, scores, cnt = t.recall_topk(q, k)
            orow, osc = o.recall_topk(tab, q, k)
            assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc)) and cnt.tolist() == [k] * R
            r1 = ctx.stats().recall_rescans
            rows2, rec2, rnk2, fus2, order2, cnt2 = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q, k)
            assert np.array_equal(rows2, orow) and np.array_equal(bits(rec2), bits(osc)) and cnt2.tolist() == [k] * R
            for r in range(0, R, 9):
                ref = o.dnn3_forward(w, pa.PREC_F32, q[r], tab[orow[r].astype(np.int64)])
                assert np.max(np.abs(rnk2[r].astype(np.float64) - ref)) <= 2e-7
                fused = o.widen_f32(rnk2[r]) * (1 + o.widen_f32(rec2[r])) ** 0.1
                assert np.max(np.abs(fus2[r] - fused)) <= 1e-12 and np.array_equal(order2[r], o.sort_scores(fus2[r], True))
            hit += (r1 > r0)
        assert hit >= 1, "the thin margin never failed: the partial fallback was not exercised"
        co = pa.Coalescer(ctx, t, k, m, ex, "gpu_dnn", max_top_n=20, max_wait_us=1000)
        q = o.synth_rows(o.SEED_QUERY, 0, 128, d)
        got = [None] * 128
        run_threads(128, lambda i: got.__setitem__(i, co.recommend(q[i], 20)))
        co.destroy()
        orow, osc = o.recall_topk(tab, q, k)
        for i in range(0, 128, 5):
            ref = o.dnn3_forward(w, pa.PREC_F32, q[i], tab[orow[i].astype(np.int64)])
            fused = o.widen_f32(ref.astype(np.float32)) * (1 + o.widen_f32(osc[i])) ** 0.1
            top = orow[i][np.argsort(-fused, kind="stable")[:20]]
            assert set(got[i][0].tolist()) <= set(orow[i].tolist())
            if np.all(np.abs(np.diff(np.sort(fused)[::-1][:21])) > 1e-6):
                assert np.array_equal(got[i][0], top)
        assert ctx.stats().recall_rescans > before
    finally:
        ctx.set_option("pilot_sigmas", 6)
        ctx.set_option("pilot_fraction", 0)
    m.destroy()
    t.destroy()


def test_recommend_pads_when_table_is_smaller_than_k(ctx):
    """Fewer rows than k: the padding slots (row = UINT64_MAX) must not surface in a page — they carry fused = NaN,
    sort last, and the count says how many entries are items (they used to fuse to +inf and sort first)."""
    n, d, k, R = 300, 128, 400, 5
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    w = o.Dnn3Weights()
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    ex = pa.Expr(EXPR)
    q = o.synth_rows(o.SEED_QUERY, 3, R, d)
    rows, rec, rnk, fus, order, cnt = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q, k)
    assert cnt.tolist() == [n] * R
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    for r in range(R):
        assert np.all(rows[r][n:] == np.uint64(2**64 - 1)) and np.all(np.isnan(fus[r][n:])) and np.all(rnk[r][n:] == 0)
        head = order[r][:n]
        assert np.all(head < n), "a padding slot sorted ahead of an item"
        assert sorted(rows[r][head].tolist()) == list(range(n))
        assert np.all(np.diff(fus[r][head]) <= 0)
        ref = o.dnn3_forward(w, pa.PREC_F32, q[r], tab[rows[r][:n].astype(np.int64)])
        assert np.max(np.abs(rnk[r][:n].astype(np.float64) - ref)) <= 2e-7
    co = pa.Coalescer(ctx, t, k, m, ex, "gpu_dnn", max_top_n=k)
    p_rows, p_rec, p_rnk, p_fus, p_cnt = co.recommend(q[0], k)
    co.destroy()
    assert p_cnt == n and np.array_equal(p_rows[:n], rows[0][order[0][:n]])
    m.destroy()
    t.destroy()


def test_concurrent_callers_throughput_is_10x_solo(ctx, world):
    """The point of the coalescer: 256 closed-loop callers get >= 10x the requests/s of one caller issuing the same
    single-request calls back to back (native load generator in libpairec_host.so; same checksum semantics)."""
    t, m, ex = world
    host = C.CDLL(os.path.join(os.path.dirname(pa.__file__), "libpairec_host.so"))

    class Res(C.Structure):
        _fields_ = [("requests", C.c_uint64), ("errors", C.c_uint64), ("seconds", C.c_double), ("p50_ms", C.c_double),
                    ("p90_ms", C.c_double), ("p99_ms", C.c_double), ("max_ms", C.c_double), ("mean_ms", C.c_double),
                    ("checksum", C.c_uint64)]
    host.ph_loadgen_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.c_double, C.POINTER(Res)]
    k, top_n = 1000, 100
    users = np.ascontiguousarray(o.synth_rows(o.SEED_QUERY, 0, 1000, 128))
    co = pa.Coalescer(ctx, t, k, m, ex, "gpu_dnn", max_top_n=top_n)
    solo, many = Res(), Res()
    assert host.ph_loadgen_run(co.h, 0, users.ctypes.data, 1000, 128, k, top_n, 1, 5, 1.0, C.byref(solo)) == 0
    assert host.ph_loadgen_run(co.h, 0, users.ctypes.data, 1000, 128, k, top_n, 256, 2, 2.0, C.byref(many)) == 0
    co.destroy()
    assert solo.errors == 0 and many.errors == 0
    rs, rm = solo.requests / solo.seconds, many.requests / many.seconds
    print("solo %.0f req/s (p50 %.2f ms), 256 callers %.0f req/s (p50 %.2f ms, p99 %.2f ms)" %
          (rs, solo.p50_ms, rm, many.p50_ms, many.p99_ms))
    assert rm >= 10 * rs


@pytest.mark.parametrize("callers", [5, 24, 48])
def test_coalesced_mid_batches_ride_the_4bit_paths_and_equal_solo_calls(ctx, world, callers):
    """Round 6: a handful to a few dozen concurrent callers make passes of 3-64 requests — the 4-bit shadow through the matrix
    pipe + its int8 stage (csrc/recall_i4m.hip; forced on for this small table), behind the rejoin hold of the coalescer.  Every
    caller's page equals the same request alone, bit for bit, and the recall flavour equals pg_recall_topk; the callers, closed
    loops, end up in ONE batch per round."""
    t, m, ex = world
    k, top_n, rounds = 500, 50, 4
    for name, v in (("i4_min_rows", "0"), ("i4m_max_pairs", "1e12"), ("i4m_max_lambda", "1000")):
        ctx.set_option(name, v)
    try:
        q = o.synth_rows(o.SEED_QUERY, 300, callers * rounds, 128)
        ref = {}
        for i in range(0, callers * rounds, 7):
            rows, rec, rnk, fus, order, cnt = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q[i:i + 1], k)
            p = order[0][:top_n]
            ref[i] = (rows[0][p], rec[0][p], rnk[0][p], fus[0][p])
        r_rows, r_sc, _ = t.recall_topk(q[:callers], k)
        co = pa.Coalescer(ctx, t, k, m, ex, "gpu_dnn", max_top_n=top_n, max_wait_us=2000)
        got = [None] * (callers * rounds)
        rec_got = [None] * callers

        def call(i):
            for r in range(rounds):
                got[i + r * callers] = co.recommend(q[i + r * callers], top_n)
            rec_got[i] = co.recall(q[i])
        run_threads(callers, call)
        st = co.stats()
        co.destroy()
        for i, (rows, rec, rnk, fus) in ref.items():
            g = got[i]
            assert g[4] == top_n and np.array_equal(g[0], rows), "request %d: page differs from the solo call" % i
            assert np.array_equal(bits(g[1]), bits(rec)) and np.array_equal(bits(g[2]), bits(rnk)) and np.array_equal(bits(g[3]), bits(fus))
        for i in range(callers):
            assert np.array_equal(rec_got[i][0], r_rows[i]) and np.array_equal(bits(rec_got[i][1]), bits(r_sc[i]))
        assert st.requests[2] == callers * rounds
        assert st.batches[2] <= rounds + 3, "the closed loops did not merge into one batch per round: %d batches" % st.batches[2]
        assert ctx.last_scan_kernel()[1] < t.rows * 128
    finally:
        for name, v in (("i4_min_rows", str(1 << 22)), ("i4m_max_pairs", "2.4e7"), ("i4m_max_lambda", "2.2")):
            ctx.set_option(name, v)


def test_coalesced_batches_record_no_stage_timers_unless_asked(ctx, world):
    """A coalesced batch records none of its context's stage-timer events (each costs ~6 us of idle queue): pg_stats' last_recall_ms /
    last_rank_ms keep what the last DIRECT call left, the pages are the same; PG_COALESCER_TIMERS=1 at creation records them again."""
    import os
    t, m, ex = world
    k, top_n = 300, 20
    q = o.synth_rows(o.SEED_QUERY, 900, 4, 128)
    rows, rec, rnk, fus, order, cnt = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q[0:1], k)
    ref = rows[0][order[0][:top_n]]
    s0 = ctx.stats()
    assert s0.last_recall_ms > 0 and s0.last_rank_ms > 0
    for env, timers in ((None, False), ("1", True)):
        if env is None:
            os.environ.pop("PG_COALESCER_TIMERS", None)
        else:
            os.environ["PG_COALESCER_TIMERS"] = env
        try:
            co = pa.Coalescer(ctx, t, k, m, ex, "gpu_dnn", max_top_n=top_n, max_wait_us=200, depth=1)
        finally:
            os.environ.pop("PG_COALESCER_TIMERS", None)
        for i in range(3):
            g = co.recommend(q[0], top_n)
            assert np.array_equal(g[0], ref)
        co.destroy()
        s1 = ctx.stats()
        unchanged = s1.last_recall_ms == s0.last_recall_ms and s1.last_rank_ms == s0.last_rank_ms
        assert unchanged == (not timers), (timers, s0.last_recall_ms, s1.last_recall_ms, s0.last_rank_ms, s1.last_rank_ms)
        s0 = s1
    # direct calls go without them on request (pg_set_option "stage_timers"): same results, the figures stop moving
    ctx.set_option("stage_timers", 0)
    try:
        rows2, rec2, rnk2, fus2, order2, cnt2 = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q[0:1], k)
        r_rows, r_sc, _ = t.recall_topk(q[1:3], k)
        s2 = ctx.stats()
        assert np.array_equal(rows2, rows) and np.array_equal(order2, order) and np.array_equal(bits(fus2), bits(fus))
        assert s2.last_recall_ms == s0.last_recall_ms and s2.last_rank_ms == s0.last_rank_ms
    finally:
        ctx.set_option("stage_timers", 1)
    pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q[0:1], k)
    s3 = ctx.stats()
    assert s3.last_recall_ms > 0 and (s3.last_recall_ms != s0.last_recall_ms or s3.last_rank_ms != s0.last_rank_ms)
