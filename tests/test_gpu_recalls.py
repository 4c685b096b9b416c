"""GPU tests of the composed recalls and of BASELINE.json's configs[0] / configs[3]
at their stated sizes, through the C ABI against the CPU oracle."""
import numpy as np
import pytest

import pairec_amd as pa
from oracle import oracle as o

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


def test_i2i_vector_recall_matches_oracle(ctx):
    """I2IVectorRecall (item_2_item_vector_racall.go:51-152): the trigger item's embedding is the query; the
    trigger itself comes back first (its own inner product is the largest for normalised rows) and is not
    excluded, as in the reference's SQL."""
    n, d, k = 120_000, 128, 50
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    trig = np.array([5, 99_999, 31_337], dtype=np.uint32)
    rows, scores, cnt = t.i2i_recall(trig, k)
    orow, osc = o.recall_topk(tab, tab[trig], k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc)) and cnt.tolist() == [k] * 3
    assert rows[:, 0].tolist() == trig.tolist()
    with pytest.raises(pa._lib.PgError):
        t.i2i_recall(np.array([n], dtype=np.uint32), k)
    t.destroy()


@pytest.mark.parametrize("prec,tol", [(pa.PREC_F32, 0.0), (pa.PREC_BF16, 0.0)])
def test_online_vector_recall_matches_oracle(ctx, prec, tol):
    """OnlineVectorRecall (online_vector_recall.go:73-155): user features → user tower → k nearest items of the
    item-embedding table.  The user tower is a sequential fmaf chain in both modes (bit-defined), so the embedding,
    the ids, the order and the scores all match the oracle exactly."""
    n, k, R = 200_000, 300, 5
    fw = o.Fm2tWeights(vocab=1000)
    m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, prec, pa.pack_fm2t(fw))
    emb = pa.Table(ctx, n, 64)                          # the item tower's outputs, as served by the vector index
    emb.fill_synthetic(o.SEED_TABLE ^ 0x77)
    tab = o.synth_rows(o.SEED_TABLE ^ 0x77, 0, n, 64)
    users = o.synth_rows(o.SEED_QUERY, 40, R, 128)
    ue = m.user_embedding(users)
    ref_ue = np.stack([o.fm2t_user_embedding(fw, prec, users[r]) for r in range(R)])
    assert np.array_equal(bits(ue), bits(ref_ue))
    rows, scores, cnt = m.online_vector_recall(emb, users, k)
    orow, osc = o.recall_topk(tab, ref_ue, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc)) and cnt.tolist() == [k] * R
    emb.destroy()
    m.destroy()


def test_cfg1_full_size_recall_and_item_score_sort(ctx):
    """BASELINE.json configs[0] at its own size: 1M x 64, top-200, then sort.item_score (ASCENDING,
    sort/item_score.go:36-41) — ids, order and score bits equal the oracle's."""
    n, d, k, R = 1_000_000, 64, 200, 16
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    q = o.synth_rows(o.SEED_QUERY, 0, R, d)
    rows, scores, cnt = t.recall_topk(q, k)
    orow, osc = o.recall_topk(tab, q, k)
    assert np.array_equal(rows, orow) and np.array_equal(bits(scores), bits(osc))
    one = t.recall_topk(q[3:4], k)                      # one request per pass gives the same answer
    assert np.array_equal(one[0][0], orow[3]) and np.array_equal(bits(one[1][0]), bits(osc[3]))
    off = (np.arange(R + 1) * k).astype(np.uint32)
    order = ctx.sort_scores(scores.reshape(-1).astype(np.float64), off, descending=False)
    for r in range(R):
        assert np.array_equal(order[off[r]:off[r + 1]], o.sort_scores(osc[r].astype(np.float64), False))
    t.destroy()


def test_cfg4_fm_twotower_with_million_row_field_tables(ctx):
    """BASELINE.json configs[3] with SURVEY.md 8d's field tables (8 + 8 fields x 1M rows x k=16): 5000 candidates
    of 4 requests, ids spread over the whole vocabulary."""
    vocab, R, K = 1_000_000, 4, 5000
    fw = o.Fm2tWeights(vocab=vocab)
    rng = np.random.default_rng(11)
    users = o.synth_rows(o.SEED_QUERY, 0, R, 128)
    ufids = rng.integers(0, vocab, (R, 8)).astype(np.int32)
    ifids = rng.integers(0, vocab, (R * K, 8)).astype(np.int32)
    ifids[0] = vocab - 1                                 # last row of every field table
    ifids[1] = 0
    off = (np.arange(R + 1) * K).astype(np.uint32)
    for prec, tol in ((pa.PREC_F32, 2e-7), (pa.PREC_BF16, 1e-5)):
        m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, prec, pa.pack_fm2t(fw))
        got = m.rank_fm2t(users, ufids, ifids, off)
        for r in range(R):
            ref = o.fm2t_forward(fw, prec, users[r], ufids[r], ifids[off[r]:off[r + 1]])
            assert np.max(np.abs(got[off[r]:off[r + 1]].astype(np.float64) - ref)) <= tol
        m.destroy()


@pytest.mark.parametrize("d_user,d_item,h1,h2", [(128, 128, 128, 128), (64, 64, 256, 128), (128, 128, 256, 256),
                                                 (200, 128, 1024, 512), (128, 64, 512, 256)])
def test_dnn3_shapes_match_oracle(ctx, d_user, d_item, h1, h2):
    """EAS-shaped DNN predict beyond the benchmark's 256-512-256-1 (algorithm/eas/model.go:197-222 serves whatever
    the model is): hidden widths, user widths and 64-wide item rows, both precision modes against the oracle."""
    n, R, K = 5000, 3, 700
    t = pa.Table(ctx, n, d_item)
    t.fill_synthetic(o.SEED_TABLE)
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d_item)
    w = o.Dnn3Weights(d_user=d_user, d_item=d_item, h1=h1, h2=h2, seed=o.SEED_WEIGHTS ^ (h1 + h2))
    users = o.synth_rows(o.SEED_QUERY, 9, R, 256)[:, :d_user].copy()
    rng = np.random.default_rng(h1)
    cand = rng.integers(0, n, R * K).astype(np.uint32)
    off = (np.arange(R + 1) * K).astype(np.uint32)
    for prec, tol in ((pa.PREC_F32, 2e-7), (pa.PREC_BF16, 1e-5)):
        m = pa.RankModel(ctx, pa.MODEL_DNN3, prec, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, d_user))
        got = m.rank_dnn3(t, users, cand, off)
        for r in range(R):
            ref = o.dnn3_forward(w, prec, users[r], tab[cand[off[r]:off[r + 1]].astype(np.int64)])
            assert np.max(np.abs(got[off[r]:off[r + 1]].astype(np.float64) - ref)) <= tol, (prec, r)
        m.destroy()
    t.destroy()
    with pytest.raises(pa._lib.PgError):
        bad = o.Dnn3Weights(d_user=128, d_item=128, h1=384, h2=256)
        pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(bad.w1, bad.b1, bad.w2, bad.b2, bad.w3, bad.b3, 128))


@pytest.mark.parametrize("nuf,nif,k,th,to", [(8, 8, 16, 256, 64), (3, 4, 32, 256, 64), (16, 16, 8, 128, 64),
                                             (5, 8, 16, 512, 128)])
def test_fm2t_shapes_match_oracle(ctx, nuf, nif, k, th, to):
    """FM + two-tower over field counts / embedding widths / tower widths (n_item_fields x k = 128)."""
    vocab, R, K = 3000, 3, 500
    fw = o.Fm2tWeights(n_user_fields=nuf, n_item_fields=nif, k=k, d_user=96, t_h1=th, t_out=to, vocab=vocab)
    rng = np.random.default_rng(k + th)
    users = o.synth_rows(o.SEED_QUERY, 1, R, 128)[:, :96].copy()
    ufids = rng.integers(0, vocab, (R, nuf)).astype(np.int32)
    ifids = rng.integers(0, vocab, (R * K, nif)).astype(np.int32)
    off = (np.arange(R + 1) * K).astype(np.uint32)
    for prec, tol in ((pa.PREC_F32, 2e-7), (pa.PREC_BF16, 1e-5)):
        m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, prec, pa.pack_fm2t(fw))
        got = m.rank_fm2t(users, ufids, ifids, off)
        for r in range(R):
            ref = o.fm2t_forward(fw, prec, users[r], ufids[r], ifids[off[r]:off[r + 1]])
            assert np.max(np.abs(got[off[r]:off[r + 1]].astype(np.float64) - ref)) <= tol, (prec, r)
        m.destroy()


@pytest.mark.parametrize("nuf,nif,k,th,to", [(8, 8, 16, 256, 64), (3, 4, 32, 256, 64), (16, 16, 8, 128, 64),
                                             (5, 8, 16, 512, 128)])
def test_fm2t_materialised_item_records_are_bit_identical(ctx, nuf, nif, k, th, to):
    """pg_fm2t_item_rows_*: one contiguous record per item (the field embeddings of its static ids + linear weights)
    instead of n_item_fields scattered gathers.  Same values, same summation order: scores equal the per-field path's
    bit for bit in BOTH precision modes, for every shape; candidates outside the store read the columns' defaults;
    after a column changes, _update re-materialises the rows."""
    vocab, n_items, R, K = 3000, 20_000, 3, 700
    fw = o.Fm2tWeights(n_user_fields=nuf, n_item_fields=nif, k=k, d_user=96, t_h1=th, t_out=to, vocab=vocab)
    rng = np.random.default_rng(k + th + 1)
    users = o.synth_rows(o.SEED_QUERY, 1, R, 128)[:, :96].copy()
    ufids = rng.integers(0, vocab, (R, nuf)).astype(np.int32)
    ids = rng.integers(0, vocab, (n_items, nif)).astype(np.int32)
    feats = pa.Features(ctx, n_items)
    cols = ["c%d" % f for f in range(nif)]
    for f, c in enumerate(cols):
        feats.set_column(c, pa.F_I32 if f % 2 else pa.F_I64, ids[:, f].astype(np.int64 if f % 2 == 0 else np.int32), default=7 + f)
    cand = rng.integers(0, n_items, R * K).astype(np.uint32)
    cand[5] = n_items + 3                                # outside the store: the defaults' record
    cand[6] = 0xFFFFFFFF
    off = (np.arange(R + 1) * K).astype(np.uint32)
    for prec, tol in ((pa.PREC_F32, 2e-7), (pa.PREC_BF16, 1e-5)):
        m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, prec, pa.pack_fm2t(fw))
        ir = pa.ItemRows(m, feats, cols)
        want = m.rank_fm2t_rows(feats, cols, users, ufids, cand, off)
        got = ir.rank(users, ufids, cand, off)
        assert np.array_equal(bits(got), bits(want)), "precision mode %d: materialised records differ from the per-field path" % prec
        ids_c = np.where(cand[:, None] < n_items, ids[np.minimum(cand, n_items - 1).astype(np.int64)],
                         np.array([7 + f for f in range(nif)], dtype=np.int32)[None, :])
        for r in range(R):
            ref = o.fm2t_forward(fw, prec, users[r], ufids[r], ids_c[off[r]:off[r + 1]])
            assert np.max(np.abs(got[off[r]:off[r + 1]].astype(np.float64) - ref)) <= tol, (prec, r)
        # a column changes: the records follow after _update
        new0 = rng.integers(0, vocab, n_items).astype(np.int64)
        feats.set_column(cols[0], pa.F_I64, new0, default=7)
        ir.update(0, n_items)
        assert np.array_equal(bits(ir.rank(users, ufids, cand, off)), bits(m.rank_fm2t_rows(feats, cols, users, ufids, cand, off)))
        feats.set_column(cols[0], pa.F_I64, ids[:, 0].astype(np.int64), default=7)
        ir.destroy()
        m.destroy()
    feats.destroy()


@pytest.mark.gpu
def test_fm2t_item_record_kernels_agree_bit_for_bit_on_ragged_batches(ctx):
    """The benchmark's shape (8 x 16 item fields, towers 256 / 64, bf16) has two kernels over the item records:
    fm2t_isw_kernel (rank_is.hip, the default: every wave a whole pipeline over 32-item tiles, X and H1 in registers) and
    fm2t_irs_kernel (rank_ir.hip, option fm2t_irs: producer / consumer waves over 64-item tiles).  Both run the
    specification's chains and the same MFMA sequences: equal to each other and to the per-field path bit for bit — on
    requests of 0, 1, 31, 32, 33, 64, 65 and thousands of candidates, fewer tiles than waves, candidates outside the store."""
    vocab, n_items = 5000, 50_000
    fw = o.Fm2tWeights(vocab=vocab)
    rng = np.random.default_rng(321)
    ids = rng.integers(0, vocab, (n_items, 8)).astype(np.int32)
    feats = pa.Features(ctx, n_items)
    cols = ["c%d" % f for f in range(8)]
    for f, c in enumerate(cols):
        feats.set_column(c, pa.F_I32, np.ascontiguousarray(ids[:, f]), default=3 + f)
    m = pa.RankModel(ctx, pa.MODEL_FM_TWOTOWER, pa.PREC_BF16, pa.pack_fm2t(fw))
    ir = pa.ItemRows(m, feats, cols)
    for sizes in ([5], [0, 1, 31, 32, 33, 64, 65, 0, 700], [4000, 1, 2500], [32] * 40):
        R = len(sizes)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
        n = int(off[-1])
        users = o.synth_rows(o.SEED_QUERY, 3, R, 128)
        ufids = rng.integers(0, vocab, (R, 8)).astype(np.int32)
        cand = rng.integers(0, n_items, n).astype(np.uint32)
        cand[0] = n_items + 1                                   # the defaults' record
        want = m.rank_fm2t_rows(feats, cols, users, ufids, cand, off)
        got = {}
        for irs in (0, 1):
            ctx.set_option("fm2t_irs", irs)
            got[irs] = ir.rank(users, ufids, cand, off)
        ctx.set_option("fm2t_irs", 0)
        assert np.array_equal(bits(got[0]), bits(want)), sizes
        assert np.array_equal(bits(got[1]), bits(want)), sizes
    ir.destroy()
    m.destroy()
    feats.destroy()
