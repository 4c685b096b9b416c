"""RankConfig.ScoreRewrite (recconf/recconf.go:743; service/rank/rank_service.go:296-306,343-353).

Per item the reference evaluates EVERY source's expression over the item as the rank algorithms left it, collects the
results in a map, writes them back with Item.AddAlgoScores (module/item.go:177-188) and only then evaluates RankScore; a
source whose expression does not compile scores 0.  Checked here: the oracle's restatement on hand-computed values (CPU);
the device's fusion stage — pg_recommend_dnn3, the scene coalescer with two models, the shard group — against the oracle
with two rewrites feeding RankScore (one of them overwriting an algorithm's own name, one introducing a new name, each
reading the OTHER's un-rewritten input); the host mirror driven by recconf JSON."""
import json

import numpy as np
import pytest

import pairec_amd as pa
from oracle import oracle as o
from test_host_mirror import H  # noqa: F401 — the mirror library fixture

REWRITES = {"gpu_dnn": "${gpu_dnn}*0.5+${current_score}*0.01", "boost": "${gpu_dnn}^2+${current_score}"}


def test_oracle_rewrites_read_the_unrewritten_item_and_failed_sources_score_zero():
    it = o.OracleItem("a", 0.25)
    it.add_algo_score("ctr", 0.5)
    it.add_algo_score("cvr", 0.125)
    # both sources read the ORIGINAL ctr / cvr, whatever the map's order; "bad" does not compile → 0
    rw = {"ctr": "${ctr}+${cvr}", "cvr": "${ctr}*4", "bad": "${ctr} @ 1", "fresh": "${current_score}*2"}
    o.fuse_scores("${ctr}*100+${cvr}*10+${bad}+${fresh}", [it], score_rewrite=rw)
    assert it.algo_scores["ctr"] == 0.625 and it.algo_scores["cvr"] == 2.0
    assert it.algo_scores["bad"] == 0.0 and it.algo_scores["fresh"] == 0.5
    assert it.score == 0.625 * 100 + 2.0 * 10 + 0.0 + 0.5
    # an empty RankScore: no rewrite either (rank_service.go:339)
    it2 = o.OracleItem("b", 1.0)
    it2.add_algo_score("ctr", 0.5)
    o.fuse_scores("", [it2], score_rewrite={"ctr": "${ctr}*2"})
    assert it2.algo_scores["ctr"] == 0.5 and it2.score == 1.0


def _oracle_fused(rank_by_name, recall, expr, rewrites):
    """fused score per candidate through the oracle: rank_by_name {name: f32 scores}, recall f32 scores."""
    out = np.empty(len(recall), np.float64)
    for i in range(len(recall)):
        it = o.OracleItem(str(i), float(recall[i]))
        for nm, sc in rank_by_name.items():
            it.add_algo_score(nm, float(np.float32(sc[i])))
        o.fuse_scores(expr, [it], score_rewrite=rewrites)
        out[i] = it.score
    return out


@pytest.mark.gpu
def test_device_fusion_with_two_rewrites_matches_the_oracle(ctx):
    n, d, k, R = 50_000, 128, 300, 3
    t = pa.Table(ctx, n, d)
    t.fill_synthetic(o.SEED_TABLE)
    w = o.Dnn3Weights()
    m = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    expr_src = "${gpu_dnn}*(1+${boost})^0.1"
    ex = pa.Expr(expr_src)
    ex.set_score_rewrites(REWRITES)
    q = o.synth_rows(o.SEED_QUERY, 40, R, d)
    rows, rec, rnk, fus, order, _ = pa.recommend_dnn3(ctx, t, m, ex, "gpu_dnn", q, k)
    for r in range(R):
        want = _oracle_fused({"gpu_dnn": rnk[r]}, rec[r], expr_src, REWRITES)
        assert np.max(np.abs(fus[r] - want) / np.maximum(np.abs(want), 1e-300)) <= 1e-12
        assert np.array_equal(order[r], o.sort_scores(fus[r], True))
    # without the rewrites the same expression cannot bind "boost"
    ex2 = pa.Expr(expr_src)
    with pytest.raises(pa._lib.PgError):
        pa.recommend_dnn3(ctx, t, m, ex2, "gpu_dnn", q, k)
    # a rewrite whose variable is unknown is refused when the pipeline binds it; a source that does not compile scores 0
    ex2.set_score_rewrites({"boost": "${nobody}+1"})
    with pytest.raises(pa._lib.PgError):
        pa.recommend_dnn3(ctx, t, m, ex2, "gpu_dnn", q, k)
    ex2.set_score_rewrites({"boost": "${gpu_dnn} @ 1"})
    r2 = pa.recommend_dnn3(ctx, t, m, ex2, "gpu_dnn", q[:1], k)
    assert np.array_equal(r2[3][0], r2[2][0].astype(np.float64))          # gpu_dnn * (1 + 0)^0.1
    # a division by zero inside a rewrite is the request's arithmetic error, like one in RankScore itself
    ex2.set_score_rewrites({"boost": "1/(${gpu_dnn}-${gpu_dnn})"})
    with pytest.raises(pa._lib.PgError) as ei:
        pa.recommend_dnn3(ctx, t, m, ex2, "gpu_dnn", q[:1], k)
    assert ei.value.code == -5
    ex2.free()

    # pg_fuse_scores_dev: the fusion alone (what pairec_amd/dist.py calls between its collectives), same answers
    import ctypes as C
    n_it = int(rnk[0].size)
    d_rank, d_rec, d_out = ctx.to_device(rnk[0]), ctx.to_device(rec[0]), ctx.malloc(n_it * 8)
    names = (C.c_char_p * 1)(b"gpu_dnn")
    pa._lib.check(ctx.L.pg_fuse_scores_dev(ctx.h, ex.h, names, 1, d_rank, n_it, d_rec, n_it, d_out))
    alone = np.zeros(n_it, np.float64)
    ctx.d2h(alone, d_out)
    assert np.array_equal(alone.view(np.uint64), fus[0].view(np.uint64))
    ex2 = pa.Expr("1/(${gpu_dnn}-${gpu_dnn})")
    with pytest.raises(pa._lib.PgError) as ei:
        pa._lib.check(ctx.L.pg_fuse_scores_dev(ctx.h, ex2.h, names, 1, d_rank, n_it, d_rec, n_it, d_out))
    assert ei.value.code == -5
    ex2.free()
    for p_ in (d_rank, d_rec, d_out):
        ctx.free(p_)

    # the scene coalescer: two algorithms, a rewrite over both feeding RankScore; single-request calls
    w2 = o.Dnn3Weights(h1=256, h2=128, seed=o.SEED_WEIGHTS ^ 0x51)
    m2 = pa.RankModel(ctx, pa.MODEL_DNN3, pa.PREC_F32, pa.pack_dnn3(w2.w1, w2.b1, w2.w2, w2.b2, w2.w3, w2.b3, 128))
    src2 = "${mix}*(1+${current_score})^0.1+${b}*0.001"
    rw2 = {"mix": "${a}*0.7+${b}*0.3", "b": "${b}*${a}"}
    ex3 = pa.Expr(src2)
    ex3.set_score_rewrites(rw2)
    co = pa.Coalescer(ctx, t, k, expr=ex3, algos=[("a", m), ("b", m2)], max_top_n=50, max_wait_us=200)
    # the coalescer holds bindings sized for these rewrites: changing them now is refused, and nothing changes
    with pytest.raises(pa._lib.PgError) as ei:
        ex3.set_score_rewrites({"mix": "${a}"})
    assert ei.value.code == -1
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    for r in range(R):
        g_rows, g_rec, g_rnk, g_fus, cnt = co.recommend(q[r], 50)
        assert cnt == 50 and g_rnk.shape == (2, 50)
        # the page's model scores are the algorithms' own (un-rewritten) planes; its fused scores obey the rewrites
        emb = tab[g_rows.astype(np.int64)]
        assert np.max(np.abs(g_rnk[0] - o.dnn3_forward(w, 0, q[r], emb))) <= 2e-7
        assert np.max(np.abs(g_rnk[1] - o.dnn3_forward(w2, 0, q[r], emb))) <= 2e-7
        want = _oracle_fused({"a": g_rnk[0], "b": g_rnk[1]}, g_rec, src2, rw2)
        assert np.max(np.abs(g_fus - want) / np.maximum(np.abs(want), 1e-300)) <= 1e-12
        assert np.all(np.diff(g_fus) <= 0)
    co.destroy()
    ex3.free()
    ex.free()
    m2.destroy()
    m.destroy()
    t.destroy()


@pytest.mark.gpu
def test_mirror_rank_service_applies_score_rewrite(H):
    """recconf JSON with RankConf.ScoreRewrite through the mirror's RankService.Rank: items carry the rewritten algo
    scores (AddAlgoScores) and Score = RankScore over them, equal to the oracle's."""
    import copy
    from test_host_mirror import CONFIG
    cfg = copy.deepcopy(CONFIG)
    expr_src = "${gpu_dnn}*(1+${boost})^0.1"
    cfg["RankConf"]["home_feed"]["RankScore"] = expr_src
    cfg["RankConf"]["home_feed"]["ScoreRewrite"] = REWRITES
    cfg["UserDefineConfs"]["pairec_gpu"]["Algorithms"][1]["Precision"] = "bf16x3"
    bad = copy.deepcopy(cfg)
    bad["UserDefineConfs"]["pairec_gpu"]["Algorithms"][1]["Precision"] = "fp8"
    assert not H.ph_engine_create(json.dumps(bad).encode()) and b"Precision" in H.ph_last_error()
    parsed = json.loads(H.ph_parse_recconf(json.dumps(cfg).encode()))
    assert parsed["rank_home_feed"]["score_rewrite"] == REWRITES
    h = H.ph_engine_create(json.dumps(cfg).encode())
    assert h, H.ph_last_error()
    n, d = 20000, 128
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    w = o.Dnn3Weights()
    user = o.synth_rows(o.SEED_QUERY, 3, 1, d)[0]
    vec = " ".join("%d:%s" % (i + 1, repr(float(v))) for i, v in enumerate(user))
    H.ph_set_user_vector(h, b"u1", vec.encode())
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    # (prec -1: the algorithm's configured "Precision" — here bf16x3, whose scores are the fp32 path's to ~1e-7)
    assert H.ph_engine_load_dnn3(h, -1, blob, len(blob)) == 0, H.ph_last_error()
    out = json.loads(H.ph_recommend(h, b"u1", 50, b"home_feed"))
    rows, scores = o.recall_topk(tab, user[None], 300)
    items = [o.OracleItem("item_%d" % r, float(s), "gpu_vector_recall") for r, s in zip(rows[0], scores[0])]
    dnn = o.dnn3_forward(w, 0, user, tab[rows[0].astype(np.int64)])
    for it, s in zip(items, dnn):
        it.add_algo_score("gpu_dnn", float(np.float32(s)))
    o.fuse_scores(expr_src, items, score_rewrite=REWRITES)
    by_id = {it.id: it for it in items}
    assert len(out["items"]) == 50
    for g in out["items"]:
        w_ = by_id[g["item_id"]]
        assert abs(g["score"] - w_.score) <= 1e-6
        assert abs(g["algo_scores"]["gpu_dnn"] - w_.algo_scores["gpu_dnn"]) <= 1e-6       # the REWRITTEN score (x 0.5 + …)
        assert abs(g["algo_scores"]["boost"] - w_.algo_scores["boost"]) <= 1e-6
    assert all(a["score"] >= b["score"] for a, b in zip(out["items"], out["items"][1:]))
    H.ph_engine_destroy(h)
