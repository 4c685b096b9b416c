"""GPU tests of the scene coalescer (pg_coalescer_create_scene): EVERY per-request plug-in call of a scene — recalls of
all three kinds, both rank model families, several rank algorithms with fusion over all of them, DPPSort inside the
coalesced recommend and as its own call — issued as single-request calls from >= 256 host threads must give answers
bit-identical to the same requests issued alone (SURVEY.md 8b "Threading"; call sites service/recall.go:129-145,
service/rank/rank_service.go:264-289, sort/sort.go:65-125, sort/dpp_sort.go:271-351), and deadlines must hold."""
import threading
import time

import numpy as np
import pytest

import pairec_amd as pa
from oracle import oracle as o

pytestmark = pytest.mark.gpu

CALLERS = 256


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def run_threads(n, fn):
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
    """A 300k x 128 item table, two DNN3 models, an FM + two-tower model over 8 item-field columns of the same rows."""
    n, d, vocab = 300_000, 128, 5000
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    w1 = o.Dnn3Weights()
    w2 = o.Dnn3Weights(h1=256, h2=128, seed=o.SEED_WEIGHTS ^ 0x51)
    m1 = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_BF16, pa.pack_dnn3(w1.w1, w1.b1, w1.w2, w1.b2, w1.w3, w1.b3, 128))
    m2 = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w2.w1, w2.b1, w2.w2, w2.b2, w2.w3, w2.b3, 128))
    fw = o.Fm2tWeights(vocab=vocab)
    fm = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, pa.PREC_BF16, pa.pack_fm2t(fw))
    feats = pa.Features(ctx, n)
    rng = np.random.default_rng(3)
    cols = ["f%d" % f for f in range(8)]
    ids = rng.integers(0, vocab, (n, 8)).astype(np.int32)
    for f, c in enumerate(cols):
        feats.set_column(c, pa.F_I32, ids[:, f])
    yield {"t": t, "m1": m1, "m2": m2, "fm": fm, "feats": feats, "cols": cols, "ids": ids, "vocab": vocab, "n": n}
    feats.destroy()
    fm.destroy()
    m2.destroy()
    m1.destroy()
    t.destroy()


def solo_page(ctx, t, m, ex, q, k, top_n, C, alpha, window, norm):
    """One request alone: pg_recommend_dnn3_dev, then DPPSort on the first C entries of its sorted list (pg_dpp_ex)."""
    rows, rec, rnk, fus, order, _ = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q[None, :], k)
    head = order[0][:C]
    try:
        picks, _ = pa.dpp_ex(ctx, t, rows[0][head].astype(np.uint32), fus[0][head], alpha, top_n, window,
                             norm_relevance_score=norm)
    except pa._lib.PgError as e:
        assert e.code == -5                      # "all item score is zero": the items stay as they are
        picks = np.arange(top_n)
    p = head[picks]
    return rows[0][p], rec[0][p], rnk[0][p], fus[0][p]


@pytest.mark.parametrize("norm", [0, 1, 2])
def test_coalesced_recommend_with_dpp_stage_equals_solo(ctx, world, norm):
    """A [GpuItemRankScore, GpuDPP] scene as ONE coalesced call per request: 256 threads, pages of different sizes in
    one batch; every page equals the solo pipeline + pg_dpp_ex — ids, pick order, every score bit."""
    t, m = world["t"], world["m1"]
    ex = pa.Expr("${gpu_dnn}*(1+${current_score})^0.1")
    k, C, alpha, window = 400, 120, 1.0, 10
    q = o.synth_rows(o.SEED_QUERY, 100 + norm, CALLERS, 128)
    tops = [25 if i % 3 else 40 for i in range(CALLERS)]
    co = pa.Coalescer(ctx, t, k, expr=ex, algos=[("gpu_dnn", m)], max_top_n=40, max_wait_us=2000,
                      dpp={"candidates": C, "alpha": alpha, "window": window, "norm_relevance_score": norm})
    got = [None] * CALLERS
    run_threads(CALLERS, lambda i: got.__setitem__(i, co.recommend(q[i], tops[i])))
    st = co.stats()
    co.destroy()
    for i in range(0, CALLERS, 17):
        rows, rec, rnk, fus = solo_page(ctx, t, m, ex, q[i], k, tops[i], C, alpha, window, norm)
        g = got[i]
        assert g[4] == tops[i]
        assert np.array_equal(g[0], rows), "request %d: page differs from the solo pipeline + DPP" % i
        assert np.array_equal(bits(g[1]), bits(rec)) and np.array_equal(bits(g[2]), bits(rnk)) and np.array_equal(bits(g[3]), bits(fus))
    assert st.requests[2] == CALLERS and st.batches[2] <= CALLERS // 8
    ex.free()


def test_coalesced_fm2t_rank_equals_direct_call(ctx, world):
    """The FM + two-tower IAlgorithm.Run as RankService.Rank issues it — one 100-item batch per call, 3 calls per
    request, 256 requests at once (rank_service.go:264-289) — against pg_rank_fm2t_rows on the same inputs."""
    t, fm, feats, cols = world["t"], world["fm"], world["feats"], world["cols"]
    rng = np.random.default_rng(8)
    users = o.synth_rows(o.SEED_QUERY, 7, CALLERS, 128)
    ufids = rng.integers(0, world["vocab"], (CALLERS, 8)).astype(np.int32)
    cand = rng.integers(0, world["n"], (CALLERS, 300)).astype(np.uint32)
    co = pa.Coalescer(ctx, t, 300, algos=[("fm", fm, feats, cols)], max_rank_items=100, max_wait_us=2000)
    got = [None] * CALLERS

    def call(i):
        got[i] = np.concatenate([co.rank_fm2t(users[i], ufids[i], cand[i][b:b + 100]) for b in range(0, 300, 100)])
    run_threads(CALLERS, call)
    st = co.stats()
    off = (np.arange(CALLERS + 1) * 300).astype(np.uint32)
    ref = fm.rank_fm2t_rows(feats, cols, users, ufids, cand.reshape(-1), off).reshape(CALLERS, 300)
    for i in range(CALLERS):
        assert np.array_equal(bits(got[i]), bits(ref[i])), "caller %d" % i
    assert st.requests[1] == 3 * CALLERS and st.batches[1] < 3 * CALLERS // 8
    # and against the oracle (the columns hold the item field ids)
    fw = o.Fm2tWeights(vocab=world["vocab"])
    r0 = o.fm2t_forward(fw, 1, users[5], ufids[5], world["ids"][cand[5].astype(np.int64)])
    assert np.max(np.abs(got[5].astype(np.float64) - r0)) <= 1e-5
    with pytest.raises(pa._lib.PgError):
        co.rank_fm2t(users[0], np.full(8, world["vocab"], np.int32), cand[0][:10])       # user field id outside the vocabulary
    co.destroy()
    # the same algorithm over its materialised item records: same bits
    ir = pa.ItemRows(fm, feats, cols)
    co = pa.Coalescer(ctx, t, 300, algos=[("fm", fm, ir)], max_rank_items=100, max_wait_us=2000)
    got2 = [None] * CALLERS
    run_threads(CALLERS, lambda i: got2.__setitem__(i, co.rank_fm2t(users[i], ufids[i], cand[i][:100])))
    co.destroy()
    ir.destroy()
    for i in range(CALLERS):
        assert np.array_equal(bits(got2[i]), bits(ref[i][:100])), "caller %d (item records)" % i


def test_coalesced_recalls_of_every_kind_share_passes(ctx, world):
    """Vector recalls, I2I recalls (trigger rows) and — on the item-embedding table of a vector model — online vector
    recalls, mixed in the same batches, equal pg_recall_topk / pg_i2i_recall / pg_online_vector_recall alone."""
    t = world["t"]
    k = 200
    q = o.synth_rows(o.SEED_QUERY, 900, CALLERS, 128)
    trig = (np.arange(CALLERS, dtype=np.uint32) * 997) % world["n"]
    co = pa.Coalescer(ctx, t, k, algos=[], max_wait_us=2000)
    got = [None] * CALLERS
    run_threads(CALLERS, lambda i: got.__setitem__(i, co.i2i_recall(trig[i]) if i % 2 else co.recall(q[i])))
    st = co.stats()
    co.destroy()
    rv, sv, _ = t.recall_topk(q, k)
    ri, si, _ = t.i2i_recall(trig[:32], k)
    for i in range(CALLERS):
        if i % 2 == 0:
            assert np.array_equal(got[i][0], rv[i]) and np.array_equal(bits(got[i][1]), bits(sv[i]))
        elif i < 32:
            assert np.array_equal(got[i][0], ri[i]) and np.array_equal(bits(got[i][1]), bits(si[i]))
            assert got[i][0][0] == trig[i]
    assert st.requests[0] == CALLERS and st.batches[0] <= CALLERS // 8
    # online vector recall: its own table (the item tower's outputs, dim 64) and the model's user tower in front
    n_e = 150_000
    emb = pa.Table(ctx, n_e, 64)
    emb.fill_synthetic(o.SEED_TABLE ^ 0x77)
    co = pa.Coalescer(ctx, emb, k, query_model=world["fm"], max_wait_us=2000)
    got = [None] * CALLERS
    run_threads(CALLERS, lambda i: got.__setitem__(i, co.online_recall(q[i]) if i % 4 else co.recall(q[i][:64])))
    co.destroy()
    ro, so, _ = world["fm"].online_vector_recall(emb, q[:24], k)
    rq, sq, _ = emb.recall_topk(q[:24, :64], k)
    for i in range(24):
        want = (ro[i], so[i]) if i % 4 else (rq[i], sq[i])
        assert np.array_equal(got[i][0], want[0]) and np.array_equal(bits(got[i][1]), bits(want[1])), i
    emb.destroy()


def test_coalesced_squared_euclidean_recalls_beside_inner_product_ones(ctx, world):
    """HologresVectorRecallV2 calls (pg_coalescer_recall_l2) and VectorRecall calls on the same scene coalescer, from 256
    threads at once: each metric has its own queue and passes (up to 32 squared-Euclidean queries per exact pass), every
    answer equals the solo call's — ids, order and score / distance bits."""
    t = world["t"]
    k = 150
    q = o.synth_rows(o.SEED_QUERY, 400, CALLERS, 128) * np.linspace(0.6, 1.4, CALLERS, dtype=np.float32)[:, None]
    co = pa.Coalescer(ctx, t, k, algos=[], max_wait_us=2000)
    got = [None] * CALLERS
    run_threads(CALLERS, lambda i: got.__setitem__(i, co.recall_l2(q[i]) if i % 3 else co.recall(q[i])))
    st = co.stats()
    co.destroy()
    sel = list(range(0, CALLERS, 7))
    rl, dl, _ = t.recall_topk_l2(q[sel], k)
    rv, sv, _ = t.recall_topk(q[sel], k)
    for n_, i in enumerate(sel):
        want = (rl[n_], dl[n_]) if i % 3 else (rv[n_], sv[n_])
        assert np.array_equal(got[i][0], want[0]) and np.array_equal(bits(got[i][1]), bits(want[1])), i
        assert got[i][2] == k
    assert st.requests[0] == CALLERS and st.batches[0] <= CALLERS // 8 + 8


def test_coalesced_recommend_with_three_rank_algorithms(ctx, world):
    """RankAlgoList with three entries (two DNNs, one FM + two-tower) and a RankScore over all of them plus
    current_score (rank_service.go:259-289,339-363): the coalesced page against the stages called one by one."""
    t, m1, m2, fm, feats, cols = world["t"], world["m1"], world["m2"], world["fm"], world["feats"], world["cols"]
    ex = pa.Expr("(${ctr}+2*${cvr})*${fm}^0.5+0.01*${current_score}")
    k, top_n = 300, 30
    rng = np.random.default_rng(21)
    q = o.synth_rows(o.SEED_QUERY, 300, CALLERS, 128)
    ufids = rng.integers(0, world["vocab"], (CALLERS, 8)).astype(np.int32)
    co = pa.Coalescer(ctx, t, k, expr=ex, algos=[("ctr", m1), ("cvr", m2), ("fm", fm, feats, cols)], max_top_n=top_n,
                      max_wait_us=2000)
    got = [None] * CALLERS
    run_threads(CALLERS, lambda i: got.__setitem__(i, co.recommend(q[i], top_n, ufids[i])))
    co.destroy()
    sel = list(range(0, CALLERS, 23))
    rows, rec, _ = t.recall_topk(q[sel], k)
    off = (np.arange(len(sel) + 1) * k).astype(np.uint32)
    cand = rows.reshape(-1).astype(np.uint32)
    s1 = m1.rank_dnn3(t, q[sel], cand, off)
    s2 = m2.rank_dnn3(t, q[sel], cand, off)
    s3 = fm.rank_fm2t_rows(feats, cols, q[sel], ufids[sel], cand, off)
    by = {"ctr": s1, "cvr": s2, "fm": s3, "current_score": rec.reshape(-1)}
    fused = ex.eval(ctx, np.stack([by[v].astype(np.float64) for v in ex.var_names]))
    order = ctx.sort_scores(fused, off, descending=True)
    for j, i in enumerate(sel):
        p = order[off[j]:off[j + 1]][:top_n]
        g = got[i]
        assert g[4] == top_n and np.array_equal(g[0], rows[j][p]), "request %d" % i
        assert np.array_equal(bits(g[1]), bits(rec[j][p])) and np.array_equal(bits(g[3]), bits(fused[off[j]:off[j + 1]][p]))
        for a, s_ in enumerate((s1, s2, s3)):
            assert np.array_equal(bits(g[2][a]), bits(s_[off[j]:off[j + 1]][p])), (i, a)
    ex.free()


def test_coalesced_dpp_calls_equal_pg_dpp_ex(ctx, world):
    """DPPSort.Sort as its own plug-in call, once per request from 256 threads (sort/sort.go:65-125): two shapes mixed
    (so two kinds of batch form), relevance normalisation modes, against pg_dpp_ex alone."""
    t = world["t"]
    rng = np.random.default_rng(77)
    co = pa.Coalescer(ctx, t, 100, algos=[], max_wait_us=3000, max_rerank_items=256)
    shapes = [(200, 30, 10, 1.0, 0), (96, 20, 5, 0.7, 1)]
    cand, rel = [], []
    for i in range(CALLERS):
        n = shapes[i % 2][0]
        cand.append(rng.choice(world["n"], n, replace=False).astype(np.uint32))
        rel.append(np.sort(rng.random(n))[::-1].copy())
    got = [None] * CALLERS

    def call(i):
        n, topn, window, alpha, norm = shapes[i % 2]
        got[i] = co.dpp(cand[i], rel[i], alpha, topn, window, norm_relevance_score=norm)
    run_threads(CALLERS, call)
    st = co.stats()
    for i in range(0, CALLERS, 9):
        n, topn, window, alpha, norm = shapes[i % 2]
        picks, used = pa.dpp_ex(ctx, t, cand[i], rel[i], alpha, topn, window, norm_relevance_score=norm)
        assert np.array_equal(got[i][0], picks), "caller %d: pick sequence differs from pg_dpp_ex" % i
        assert np.array_equal(bits(got[i][1]), bits(used))
    assert st.requests[3] == CALLERS and st.batches[3] <= CALLERS // 8
    with pytest.raises(pa._lib.PgError):
        co.dpp(cand[0][:10], np.zeros(10), 1.0, 5, 10, norm_relevance_score=1)      # "all item score is zero"
    with pytest.raises(pa._lib.PgError):
        co.dpp(np.arange(300, dtype=np.uint32), np.ones(300), 1.0, 5, 10)           # beyond max_rerank_items
    co.destroy()


def test_coalesced_ssd_calls_equal_pg_ssd(ctx, world):
    """SSDSort.Sort once per request from 256 threads (sort/ssd_sort.go:110-343): two shapes mixed, quality-score
    normalisation, against pg_ssd alone — the batched launch gives every request its own workgroups and barrier."""
    t = world["t"]
    rng = np.random.default_rng(78)
    co = pa.Coalescer(ctx, t, 100, algos=[], max_wait_us=3000, max_rerank_items=512)
    shapes = [(300, 40, 5, 0.25, 0, False), (130, 25, 8, 0.4, 1, True)]
    cand, rel = [], []
    for i in range(CALLERS):
        n = shapes[i % 2][0]
        cand.append(rng.choice(world["n"], n, replace=False).astype(np.uint32))
        rel.append(np.sort(rng.random(n))[::-1].copy())
    got = [None] * CALLERS

    def call(i):
        n, topn, window, gamma, norm, star = shapes[i % 2]
        got[i] = co.ssd(cand[i], rel[i], gamma, topn, window, norm_quality_score=norm, use_ssd_star=star)
    run_threads(CALLERS, call)
    st = co.stats()
    for i in range(0, CALLERS, 7):
        n, topn, window, gamma, norm, star = shapes[i % 2]
        picks, qual = pa.ssd(ctx, t, cand[i], rel[i], gamma, topn, window, norm_quality_score=norm, use_ssd_star=star)
        assert np.array_equal(got[i][0], picks), "caller %d: pick sequence differs from pg_ssd" % i
        assert np.array_equal(bits(got[i][1]), bits(qual))
    assert st.requests[4] == CALLERS and st.batches[4] <= CALLERS // 8
    with pytest.raises(pa._lib.PgError):
        co.ssd(cand[0], rel[0], 0.25, 10, 20)              # window > 16: not a batchable shape
    co.destroy()


def test_deadline_returns_timeout_and_later_calls_succeed(ctx, world):
    """timeout_us (eas/client.go:53-58): with the stream stalled, every caller returns PG_ERR_TIMEOUT within the bound —
    those whose batch is on the device and those still queued — and once the stall is over the same coalescer serves
    requests again, with correct answers."""
    t = world["t"]
    k = 100
    q = o.synth_rows(o.SEED_QUERY, 1234, 16, 128)
    co = pa.Coalescer(ctx, t, k, algos=[], depth=1, max_wait_us=200, timeout_us=150_000)
    rows, sc, cnt = co.recall(q[0])                  # warm: the shadow is built, buffers exist
    ctx.debug_stall(1200)                            # the coalescer's (only) stream is busy for 1.2 s
    t0 = time.perf_counter()
    codes, waits = [None] * 16, [None] * 16

    def call(i):
        t1 = time.perf_counter()
        try:
            co.recall(q[i])
            codes[i] = 0
        except pa._lib.PgError as e:
            codes[i] = e.code
        waits[i] = time.perf_counter() - t1
    run_threads(16, call)
    assert codes == [-7] * 16, codes                 # PG_ERR_TIMEOUT
    assert max(waits) < 0.6, waits                   # well before the stall ends
    assert co.stats().timeouts == 16
    time.sleep(max(0.0, 1.4 - (time.perf_counter() - t0)))
    r2, s2, c2 = co.recall(q[0])
    assert np.array_equal(r2, rows) and np.array_equal(bits(s2), bits(sc)) and c2 == k
    co.destroy()
    # the blocking pipeline call with a deadline: PG_ERR_TIMEOUT leaves the ticket valid
    import ctypes as C
    m = world["m1"]
    ex = pa.Expr("${gpu_dnn}")
    d_q = ctx.to_device(q[:2])
    n = 2 * k
    bufs = [ctx.malloc(n * 8), ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 8), ctx.malloc(n * 4), ctx.malloc(16)]
    tk = C.c_void_p()
    ctx.debug_stall(400)
    pa._lib.check(ctx.L.pg_recommend_dnn3_begin(ctx.h, t.h, m.h, ex.h, b"gpu_dnn", d_q, 2, k, *bufs, C.byref(tk)))
    assert ctx.L.pg_recommend_end_timed(ctx.h, tk, 50_000, None) == -7
    assert ctx.L.pg_recommend_end_timed(ctx.h, tk, 5_000_000, None) == 0
    out = np.zeros((2, k), np.uint64)
    ctx.d2h(out, bufs[0])
    assert np.array_equal(out[0], rows)
    for p in [d_q] + bufs:
        ctx.free(p)
    ex.free()
