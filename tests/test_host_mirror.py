"""Tests of the C++ host mirror (pairec_amd/host): registries, recconf subset, UniqueFilter and the
vector text format on CPU; the config-driven recall → rank → sort pipeline on the GPU."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    # (PH_HOST_LIB: scripts/host_asan.sh / host_tsan.sh point the same tests at the sanitizer builds of the library)
    L = C.CDLL(os.environ.get("PH_HOST_LIB") or os.path.join(ROOT, "pairec_amd", "libpairec_host.so"))
    L.ph_last_error.restype = C.c_char_p
    L.ph_engine_create.restype = C.c_void_p
    L.ph_engine_create.argtypes = [C.c_char_p]
    L.ph_engine_destroy.argtypes = [C.c_void_p]
    L.ph_engine_load_dnn3.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.ph_set_user_vector.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.ph_recommend.restype = C.c_char_p
    L.ph_recommend.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p]
    L.ph_parse_vector_string.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int]
    L.ph_unique_filter.restype = C.c_char_p
    L.ph_unique_filter.argtypes = [C.c_char_p]
    L.ph_parse_recconf.restype = C.c_char_p
    L.ph_parse_recconf.argtypes = [C.c_char_p]
    return L


RANK_SCORE = "${gpu_dnn}*(1+${current_score})^0.1"
CONFIG = {
    "RunMode": "product",
    "AlgoConfs": [],
    "RecallConfs": [],
    "SceneConfs": {"home_feed": {"default": {"RecallNames": ["gpu_vector_recall"]}}},
    "RankConf": {"home_feed": {"RankAlgoList": ["gpu_dnn"], "RankScore": RANK_SCORE, "BatchCount": 100}},
    "SortNames": {"home_feed": ["ItemRankScore"]},
    "UserDefineConfs": {"pairec_gpu": {"Device": 0,
                                       "Table": {"Rows": 20000, "Dim": 128, "IdPrefix": "item_",
                                                 "SyntheticSeed": o.SEED_TABLE},
                                       "Recalls": [{"Name": "gpu_vector_recall", "Kind": "vector", "RecallCount": 300,
                                                    "RecallAlgo": "gpu_faiss", "ItemType": "video"}],
                                       "Algorithms": [{"Name": "gpu_faiss", "Kind": "faiss"},
                                                      {"Name": "gpu_dnn", "Kind": "dnn3"}]}},
}


def test_registry_semantics(H):
    # sort: first registration wins / nil rejected; algorithm: overwrite / unknown name errors;
    # recall: unknown name errors  (sort.go:143-150, algorithm.go:107-120,164-168, recall.go:35-45)
    assert H.ph_registry_semantics() == 31


def test_gpu_recall_where_clause_forms(H):
    """HologresVectorConf.WhereClause restricts the reference's SQL candidates (hologres_vector_recall.go:49-62).  The device
    serves `column OP integer` (with ${time}) on the Kinds hologres / hologres_v2; any other clause, or a clause on another
    Kind, is an error at load time — not a silent answer over other candidates."""
    import copy
    cfg = copy.deepcopy(CONFIG)
    rec = cfg["UserDefineConfs"]["pairec_gpu"]["Recalls"]
    rec[0]["WhereClause"] = "create_time > ${time}"                     # Kind "vector" (an IAlgorithm's search): no filter there
    assert not H.ph_parse_recconf(json.dumps(cfg).encode())
    assert b"WhereClause is not supported by Kind" in H.ph_last_error()
    assert not H.ph_engine_create(json.dumps(cfg).encode()) and b"WhereClause" in H.ph_last_error()
    del rec[0]["WhereClause"]
    rec.append({"Name": "holo", "Kind": "hologres", "RecallCount": 50, "HologresVectorConf": {"WhereClause": "create_time > ${time}", "TimeInterval": 3600}})
    assert H.ph_parse_recconf(json.dumps(cfg).encode()), H.ph_last_error()
    for ok in ("stock>=5", " cat_id = 17 ", "cat_id <> -3", "ts<=1700000000", "a_b1 != 0", "x == 2", "x < ${time}"):
        rec[1]["HologresVectorConf"]["WhereClause"] = ok
        assert H.ph_parse_recconf(json.dumps(cfg).encode()), (ok, H.ph_last_error())
    for bad in ("create_time > ${time} and stock > 0", "lower(name) = 'x'", "price > 1.5", "cat in (1,2)", "> 5", "1x > 5", "x >", "x > 5 5",
                "x > 99999999999999999999"):
        rec[1]["HologresVectorConf"]["WhereClause"] = bad
        assert not H.ph_parse_recconf(json.dumps(cfg).encode()), bad
        assert b"is not supported (the device serves" in H.ph_last_error()


def test_hostile_nesting_is_an_error_not_a_stack_overflow(H):
    """The mirror's JSON parser and the RankScore compiler recurse on the C++ stack: a config of a million '[' or an
    expression of a hundred thousand terms must come back as an error (encoding/json has its own "exceeded max depth")."""
    import pairec_amd as pa
    for opener in (b"[", b'{"a":'):
        assert H.ph_parse_recconf(opener * 1_000_000) is None and b"json" in H.ph_last_error()
    deep_ok = b"[" * 900 + b"]" * 900
    assert H.ph_parse_recconf(deep_ok) is None or True          # (well-formed nesting inside the bound parses; it is not a recconf)
    for src in ("1+" * 100_000 + "1", "(" * 1_000_000 + "1" + ")" * 1_000_000, "2^" * 100_000 + "2"):
        with pytest.raises(pa._lib.PgError) as ei:
            pa.Expr(src)
        assert "too large" in str(ei.value)
    e = pa.Expr("(" * 200 + "${x}" + ")" * 200 + "+1")           # deep but small: fine
    assert e.var_names == ["x"]
    e.free()


def test_parse_vector_string(H):
    buf = (C.c_float * 16)()
    text = "1:0.12 2:-0.3 junk 3:1e-2 4:x 5:1:2"
    n = H.ph_parse_vector_string(text.encode(), buf, 16)
    want = o.parse_vector_string(text)
    assert n == len(want) and list(buf)[:n] == want.tolist()


def test_unique_filter_matches_reference_semantics(H):
    items = [{"id": "1", "score": 0.5, "retrieve_id": "r1", "algo_scores": {}},
             {"id": "2", "score": 0.4, "retrieve_id": "r1", "algo_scores": {}},
             {"id": "1", "score": 0.9, "retrieve_id": "r2", "algo_scores": {"m": 0.7}}]
    out = json.loads(H.ph_unique_filter(json.dumps(items).encode()))
    oi = [o.OracleItem(i["id"], i["score"], i["retrieve_id"]) for i in items]
    oi[2].add_algo_score("m", 0.7)
    want = o.unique_filter(oi)
    assert [x["id"] for x in out] == [x.id for x in want]
    assert out[0]["recall_scores"] == want[0].recall_scores and out[0]["algo_scores"] == want[0].algo_scores


def test_recconf_subset(H):
    got = json.loads(H.ph_parse_recconf(json.dumps(CONFIG).encode()))
    assert got["recalls"] == 0 and got["gpu_recall0"] == {"name": "gpu_vector_recall", "count": 300, "algo": "gpu_faiss"}
    assert got["rank_home_feed"] == {"batch": 100, "score": RANK_SCORE, "algos": 1, "score_rewrite": {}}
    assert H.ph_parse_recconf(b"{not json") is None and b"json" in H.ph_last_error()


def test_recall_and_sort_factories_reject_what_the_reference_rejects(H):
    """recall.Load / RegisterSortWithConfig outcomes (service/recall/recall.go:47-107, sort/sort.go:162-200) for
    RecallConfs / SortConfs entries: a GPU plug-in cannot be declared there.  The round-1 INTEGRATION config
    ("RecallType": "UserCustomRecall" without DaoConf) panics a real pairec inside runBeforeStart
    (module/user_custom_recall_dao.go:12-28) — the mirror must refuse it with the same message, before any GPU work."""
    H.ph_check_recall_conf.restype = C.c_char_p
    H.ph_check_recall_conf.argtypes = [C.c_char_p]

    def check(conf):
        return H.ph_check_recall_conf(json.dumps(conf).encode()).decode()
    assert check({"Name": "r", "RecallType": "UserCustomRecall"}) == "panic: not found UserCustomRecallDao implement"
    assert check({"Name": "r", "RecallType": "VectorRecall"}) == "panic: not found VectorDao implement"
    assert check({"Name": "r", "RecallType": "GpuVectorRecall"}) == "panic: recall empty, name:r"
    assert check({"Name": "r", "RecallType": "MilvusVectorRecall"}) == "panic: recall empty, name:r"     # constructor commented out
    assert check({"Name": "r", "RecallType": "I2IVectorRecall", "VectorDaoConf": {"HologresName": "holo"}}) == \
        "panic: Postgres not found, name:holo"
    assert check({"Name": "r", "RecallType": "VectorRecall", "DaoConf": {"AdapterType": "redis"}}).startswith("unavailable: ")
    assert check({"Name": "r", "RecallType": "MockRecall"}) == "built"
    assert check({"Name": "r", "RecallType": "OnlineVectorRecall", "RecallAlgo": "x"}) == "built"
    # whole configs: the refusal happens before the engine touches a GPU, so this runs on the CPU box too
    import copy
    bad = copy.deepcopy(CONFIG)
    bad["RecallConfs"] = [{"Name": "gpu_vector_recall", "RecallType": "UserCustomRecall", "RecallCount": 300,
                           "RecallAlgo": "gpu_faiss", "ItemType": "video"}]
    assert not H.ph_engine_create(json.dumps(bad).encode())
    assert H.ph_last_error() == b"panic: not found UserCustomRecallDao implement"
    bad = copy.deepcopy(CONFIG)
    bad["SortConfs"] = [{"Name": "my_ssd", "SortType": "SSDSort", "SSDConf": {"Gamma": 0.3}}]
    assert not H.ph_engine_create(json.dumps(bad).encode()) and H.ph_last_error().startswith(b"panic: Postgres not found, name:")
    bad["SortConfs"] = [{"Name": "gpu_sort", "SortType": "GpuItemRankScore"}]
    assert not H.ph_engine_create(json.dumps(bad).encode()) and H.ph_last_error() == b"panic: Sort is nil, name:gpu_sort"


def test_integration_md_config_passes_the_factory_rules(H):
    """The recconf JSON printed in INTEGRATION.md §1 is the one a maintainer copies: it must survive the reference's
    recall.Load / RegisterSortWithConfig rules (no GPU plug-in in RecallConfs / SortConfs) and name every plug-in it
    uses under UserDefineConfs.pairec_gpu."""
    import re
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```json\n(\{.*?\n\})\n```", text, re.S)
    assert m, "INTEGRATION.md lost its recconf example"
    cfg = json.loads(m.group(1))
    H.ph_check_recall_conf.restype = C.c_char_p
    H.ph_check_recall_conf.argtypes = [C.c_char_p]
    for rc in cfg.get("RecallConfs", []):
        assert H.ph_check_recall_conf(json.dumps(rc).encode()).decode() == "built", rc
    for sc in cfg.get("SortConfs", []):
        assert sc["SortType"] not in ("DPPSort", "SSDSort")
    gpu = cfg["UserDefineConfs"]["pairec_gpu"]
    names = {r["Name"] for r in gpu["Recalls"]} | {r["Name"] for r in cfg.get("RecallConfs", [])}
    for scene in cfg["SceneConfs"].values():
        for cat in scene.values():
            assert set(cat["RecallNames"]) <= names
    algos = {a["Name"] for a in gpu["Algorithms"]}
    assert all(r["RecallAlgo"] in algos for r in gpu["Recalls"] + cfg.get("RecallConfs", []))
    assert all(set(rc["RankAlgoList"]) <= algos for rc in cfg["RankConf"].values())
    sorts = {s_["Name"] for s_ in gpu.get("Sorts", [])} | {"ItemRankScore"}
    assert all(set(v) <= sorts for v in cfg["SortNames"].values())
    got = json.loads(H.ph_parse_recconf(json.dumps(cfg).encode()))
    assert got["gpu_recalls"] == len(gpu["Recalls"]) and got["gpu_sorts"] == len(gpu.get("Sorts", []))


def test_response_decoders_reference_fixtures(H):
    """The ResponseFunc family in the mirror against the reference's own test data: easyrec_response_test.go:11-73
    (float32 widening, [N] and [1,6]-shaped outputs sliced per item), easyrec_response.go:220-236 (missing id → 0),
    :35-70 (multi-output maps, missing id → zeros, size mismatch error), fm_response.go:28-34 (label 0 → 1 - score),
    tfserving/response.go:51-64 (row-major flatten)."""
    H.ph_decode_response.restype = C.c_char_p
    H.ph_decode_response.argtypes = [C.c_char_p]

    def dec(spec):
        r = H.ph_decode_response(json.dumps(spec).encode())
        return None if r is None else json.loads(r)
    r = dec({"func": "easyrecResponseFunc", "item_ids": ["a", "missing", "b"], "results": {"a": [0.25, 9], "b": [0.75]}})
    assert [x["score"] for x in r] == [0.25, 0.0, 0.75] and not any(x["module_type"] for x in r)
    r = dec({"func": "easyrecMutValResponseFunc", "item_ids": ["a", "zz"], "outputs": ["probs_ctr", "probs_cvr"],
             "results": {"a": [0.11173942685127258, 0.006906657014042139]}})
    assert r[0]["module_type"] and r[0]["score_map"] == {"probs_ctr": 0.11173942685127258, "probs_cvr": 0.006906657014042139}
    assert r[1]["score_map"] == {"probs_ctr": 0.0, "probs_cvr": 0.0}
    assert dec({"func": "easyrecMutValResponseFunc", "item_ids": ["a"], "outputs": ["x"], "results": {"a": [1, 2]}}) is None
    assert H.ph_last_error() == b"outputs size is not equal scores"
    # easyrec_response_test.go "test multi item": per-item scalars and a [1, 6] output sliced per item, float32 widened
    f32 = lambda v: float(np.float32(v))      # noqa: E731
    r = dec({"func": "easyrecMutClassificationResponseFunc", "item_ids": ["item_1", "item_2"],
             "tf_outputs": {"probs_is_complete_play": {"float_val": [0.7, 0.3], "shape": [1]},
                            "probs_is_play_label": {"float_val": [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.11, 0.22, 0.33, 0.44, 0.55, 0.66],
                                                    "shape": [1, 6]}}})
    assert r[1]["classify"]["probs_is_complete_play"] == [f32(0.3)]
    assert len(r[1]["classify"]["probs_is_play_label"]) == 6 and r[1]["classify"]["probs_is_play_label"][5] == f32(0.66)
    assert r[0]["classify"]["probs_is_play_label"] == [f32(v) for v in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6)]
    r = dec({"func": "alinkFMResponseFunc", "predictions": [{"prediction_result": 0, "prediction_score": 0.8},
                                                            {"prediction_result": 1, "prediction_score": 0.8}]})
    assert [x["score"] for x in r] == [1 - 0.8, 0.8] == [o.alink_fm_score(0, 0.8), o.alink_fm_score(1, 0.8)]
    r = dec({"func": "tfservingResponseFunc", "tf_rows": [[0.1, 0.2], [0.3], []]})
    assert [x["score"] for x in r] == [0.1, 0.2, 0.3]
    r = dec({"func": "widenF32", "float_val": [0.8612537, 1e-7]})
    assert [x["score"] for x in r] == o.widen_f32(np.array([0.8612537, 1e-7], dtype=np.float32)).tolist()
    # round 5: the tf / torchrec / pssmart families (algorithm/eas/client.go:60-108 names them; the decoders' arithmetic is
    # float32 → float64 widening plus indexing).  outputs_list keeps the message's order: tfResponseFunc reads the FIRST output.
    fv = [0.8612537, 0.25, 1e-7]
    wide = o.widen_f32(np.array(fv, dtype=np.float32)).tolist()
    outs = [{"name": "probs_ctr", "dtype": "float", "values": fv, "shape": [3]},
            {"name": "probs_cvr", "dtype": "double", "values": [0.5, 0.125, 0.75], "shape": [3]}]
    r = dec({"func": "tfResponseFunc", "outputs_list": outs})                              # eas/tf_response.go:50-62
    assert [x["score"] for x in r] == wide and not any(x["module_type"] for x in r)
    r = dec({"func": "tfMutValResponseFunc", "outputs_list": [outs[0], {"name": "y", "dtype": "float", "values": [0.5, 0.5, 0.5], "shape": [3]}]})
    assert all(x["module_type"] for x in r) and [x["score_map"]["probs_ctr"] for x in r] == wide and r[2]["score_map"]["y"] == 0.5
    for fn in ("torchrecMutValResponseFunc", "torchrecMutValResponseFuncDebug"):            # eas/easyrec_response.go:468-535
        r = dec({"func": fn, "item_ids": ["a", "b", "c"], "outputs_list": outs})
        assert [x["score_map"]["probs_ctr"] for x in r] == wide                             # DT_FLOAT widened
        assert [x["score_map"]["probs_cvr"] for x in r] == [0.5, 0.125, 0.75] and all(x["module_type"] for x in r)   # DT_DOUBLE as is
    assert dec({"func": "torchrecMutValResponseFunc", "item_ids": ["a", "b", "c", "d"], "outputs_list": outs}) is None
    cls = [{"name": "p1", "dtype": "float", "values": [0.7, 0.3], "shape": [2]},
           {"name": "p6", "dtype": "float", "values": [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.11, 0.22, 0.33, 0.44, 0.55, 0.66], "shape": [2, 6]}]
    for fn in ("torchrecMutClassificationResponseFunc", "torchrecMutClassificationResponseFuncDebug"):     # :537-626
        r = dec({"func": fn, "item_ids": ["item_1", "item_2"], "outputs_list": cls})
        assert r[1]["classify"]["p1"] == [f32(0.3)] and r[1]["classify"]["p6"] == [f32(v) for v in (0.11, 0.22, 0.33, 0.44, 0.55, 0.66)]
        assert not r[0]["module_type"]
    r = dec({"func": "torchrecEmbeddingItemsResponseFunc", "item_ids": ["i9", "i3"],                     # :700-734
             "outputs_list": [{"name": "match_item_scores", "dtype": "float", "values": [0.8612537, 0.25], "shape": [1, 2]}]})
    assert r == [{"item_id": "i9", "score": wide[0]}, {"item_id": "i3", "score": 0.25}]
    r = dec({"func": "easyrecResponseFuncDebug", "item_ids": ["a", "missing"], "results": {"a": [0.25]}})
    assert [x["score"] for x in r] == [0.25, 0.0]
    r = dec({"func": "easyrecMutValResponseFuncDebug", "item_ids": ["a"], "outputs": ["x", "y"], "results": {"a": [0.5, 0.25]}})
    assert r[0]["score_map"] == {"x": 0.5, "y": 0.25}
    r = dec({"func": "pssmartResponseFunc", "predictions": [{"score": 0.8, "lable": "0"}, {"score": 0.8, "label": "0"},   # pmml_response.go:10-32
                                                            {"score": 0.8, "lable": "1", "label": "1"}, {"score": 0.8}]})
    assert [x["score"] for x in r] == [1 - 0.8, 1 - 0.8, 0.8, 0.8]
    assert dec({"func": "lincubResponseFunc"}) is None and b"unknown decoder" in H.ph_last_error()


@pytest.mark.gpu
def test_config_driven_pipeline_matches_oracle(H):
    """recconf JSON → recall (VectorRecall shape) → UniqueFilter → RankService (batches of 100) →
    RankScore fusion → ItemRankScore sort → items[:size], against the oracle end to end."""
    import pairec_amd as pa
    h = H.ph_engine_create(json.dumps(CONFIG).encode())
    assert h, H.ph_last_error()
    n, d = 20000, 128
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    w = o.Dnn3Weights()
    user = o.synth_rows(o.SEED_QUERY, 3, 1, d)[0]
    vec = " ".join("%d:%s" % (i + 1, repr(float(v))) for i, v in enumerate(user))
    H.ph_set_user_vector(h, b"u1", vec.encode())
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    assert H.ph_engine_load_dnn3(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    out = json.loads(H.ph_recommend(h, b"u1", 50, b"home_feed"))
    # the same request through the oracle
    rows, scores = o.recall_topk(tab, user[None], 300)
    items = [o.OracleItem("item_%d" % r, float(s), "gpu_vector_recall") for r, s in zip(rows[0], scores[0])]
    items = o.unique_filter(items)
    dnn = o.dnn3_forward(w, 0, user, tab[rows[0].astype(np.int64)])
    for it, s in zip(items, dnn):
        it.add_algo_score("gpu_dnn", float(np.float32(s)))
    o.fuse_scores(RANK_SCORE, items)
    order = o.sort_scores([it.score for it in items], True)[:50]
    want = [items[i] for i in order]
    got_ids = [x["item_id"] for x in out["items"]]
    assert len(got_ids) == 50
    # ids/order exact wherever the oracle's own scores are separated by more than the sigmoid tolerance
    for g, w_ in zip(out["items"], want):
        assert abs(g["score"] - w_.score) <= 1e-6
    sep = np.abs(np.diff([w_.score for w_ in want])) > 1e-6
    for i, (g, w_) in enumerate(zip(out["items"], want)):
        if (i == 0 or sep[i - 1]) and (i == len(want) - 1 or sep[i]):
            assert g["item_id"] == w_.id and g["retrieve_id"] == "gpu_vector_recall"
            assert abs(g["algo_scores"]["gpu_dnn"] - w_.algo_scores["gpu_dnn"]) <= 2e-7
            assert g["algo_scores"]["recall_score"] == w_.algo_scores["recall_score"]   # current_score side effect
    assert sorted(got_ids) == sorted(w_.id for w_ in want) or sep.all() is False
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_scene_feature_transforms_feed_the_rank_score(H):
    """FeatureConfs / UserFeatureConfs (recconf.go:52-53) through the request: UserFeatureService before the recalls
    (user_recommend.go:64), FeatureService between the filter and the rank call (:129), and a transformed item property as a
    RankScore variable (Item.FloatExprData falls back to Properties, item.go:189-212)."""
    import copy
    import pairec_amd as pa
    _bind_row2(H)
    cfg = copy.deepcopy(CONFIG)
    score = RANK_SCORE + " + ${boost} + ${wk}"
    cfg["RankConf"]["home_feed"]["RankScore"] = score
    cfg["UserFeatureConfs"] = {"home_feed": {"FeatureLoadConfs": [{"Features": [
        {"FeatureType": "new_feature", "FeatureStore": "user", "FeatureName": "vip2", "Normalizer": "expression", "Expression": "vip * 2"}]}]}}
    cfg["FeatureConfs"] = {"home_feed": {"FeatureLoadConfs": [{"FeatureDaoConf": {"AdapterType": "hologres"}, "Features": [
        {"FeatureType": "new_feature", "FeatureStore": "item", "FeatureName": "boost", "Normalizer": "expr",
         "Expression": "item.recall_name == 'gpu_vector_recall' ? user.vip2 * 0.25 : 100"},
        {"FeatureType": "new_feature", "FeatureStore": "item", "FeatureName": "wk", "Normalizer": "expression", "Expression": "1 / 8"},
        {"FeatureType": "raw_feature", "FeatureStore": "item", "FeatureName": "u_vip", "FeatureSource": "user:vip2"}]}]}}
    echo = json.loads(H.ph_parse_recconf(json.dumps(cfg).encode()))
    assert echo["feature_transforms"] == 3 and echo["user_feature_transforms"] == 1
    h = H.ph_engine_create(json.dumps(cfg).encode())
    assert h, H.ph_last_error()
    n, d = 20000, 128
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    w = o.Dnn3Weights()
    user = o.synth_rows(o.SEED_QUERY, 9, 1, d)[0]
    H.ph_set_user_vector(h, b"u1", " ".join("%d:%s" % (i + 1, repr(float(v))) for i, v in enumerate(user)).encode())
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    assert H.ph_engine_load_dnn3(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    r = H.ph_recommend_ab(h, b"u1", 40, b"home_feed", json.dumps({"_user_features": {"vip": 3}}).encode())
    assert r, H.ph_last_error()
    out = json.loads(r)
    rows, scores = o.recall_topk(tab, user[None], 300)
    items = [o.OracleItem("item_%d" % r_, float(s), "gpu_vector_recall") for r_, s in zip(rows[0], scores[0])]
    dnn = o.dnn3_forward(w, 0, user, tab[rows[0].astype(np.int64)])
    for it, s in zip(items, dnn):
        it.add_algo_score("gpu_dnn", float(np.float32(s)))
        it.properties = {"boost": 1.5, "wk": 0.125}                                 # vip 3 → vip2 6.0 → 6 * 0.25
    o.fuse_scores(score, items)
    order = o.sort_scores([it.score for it in items], True)[:40]
    for g, i in zip(out["items"], order):
        assert abs(g["score"] - items[i].score) <= 1e-6
    assert len(out["items"]) == 40 and min(x["score"] for x in out["items"]) > 1.6
    H.ph_engine_destroy(h)
    # a transform outside the subset stops the load, named by scene
    cfg["FeatureConfs"]["home_feed"]["FeatureLoadConfs"][0]["Features"][0]["Expression"] = "item.tags | len()"
    assert not H.ph_engine_create(json.dumps(cfg).encode())
    assert b"FeatureConfs[home_feed]" in H.ph_last_error() and b"'|'" in H.ph_last_error()


@pytest.mark.gpu
def test_custom_field_sort_reference_known_answers(H):
    """CustomFieldSort (sort/custom_field_sort.go:21-73) registered from SortConfs and from pairec_gpu.Sorts: the orders
    custom_field_sort_test.go expects, a missing field falling back to the item's own Score (:52-62), and a random case against
    numpy's stable order."""
    import copy
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))["sort_custom_field"]
    H.ph_engine_sort.restype = C.c_char_p
    H.ph_engine_sort.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    cfg = copy.deepcopy(CONFIG)
    cfg["SortConfs"] = [dict(Name="cf%d" % i, SortType="CustomFieldSort", **g["custom_field"]) for i, g in enumerate(gold)]
    cfg["UserDefineConfs"]["pairec_gpu"]["Sorts"] = [{"Name": "by_ctr_asc", "SortType": "CustomFieldSort", "SortByField": "ctr", "SortOrder": "asc"}]
    h = H.ph_engine_create(json.dumps(cfg).encode())
    assert h, H.ph_last_error()
    for i, g in enumerate(gold):
        items = [{"id": it["id"], "score": it.get("score", 0.0), "properties": {k: v for k, v in it.items() if k not in ("id", "score")}}
                 for it in g["items"]]
        got = json.loads(H.ph_engine_sort(h, b"cf%d" % i, json.dumps(items).encode(), 10))
        assert got == g["expect_ids"], g["ref"]
    # an algo score wins over a property of the same name (item.go:198-203); items without the field sort by their Score
    items = [{"id": "a", "score": 0.5, "algo_scores": {"ctr": 0.9}, "properties": {"ctr": 0.0}},
             {"id": "b", "score": 0.7, "properties": {"ctr": 0.2}},
             {"id": "c", "score": 0.6},
             {"id": "d", "score": 0.1, "properties": {"ctr": "0.65"}}]
    assert json.loads(H.ph_engine_sort(h, b"by_ctr_asc", json.dumps(items).encode(), 10)) == ["b", "c", "d", "a"]
    rng = np.random.default_rng(5)
    v = np.round(rng.random(3000), 2)                                                  # many ties
    items = [{"id": "i%d" % i, "score": 0.0, "properties": {"ctr": float(x)}} for i, x in enumerate(v)]
    got = json.loads(H.ph_engine_sort(h, b"by_ctr_asc", json.dumps(items).encode(), 10))
    assert got == ["i%d" % i for i in np.argsort(v, kind="stable")]
    assert H.ph_engine_sort(h, b"nope", b"[]", 10) is None and b"Sort:not find" in H.ph_last_error()
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_config_driven_ssd_sort_matches_oracle(H):
    """pairec_gpu.Sorts → GpuSSDSort registered by name (the reference's own SSDSort in SortConfs needs a Hologres
    datasource) with SSDSortConfig's fields (ssd_sort.go:110-343): the page is the oracle's SSD pick sequence over
    the rank-ordered candidates."""
    import copy
    import pairec_amd as pa
    cfg = copy.deepcopy(CONFIG)
    cfg["UserDefineConfs"]["pairec_gpu"]["Sorts"] = [{"Name": "my_ssd", "SortType": "SSDSort",
                                                      "SSDConf": {"Gamma": 0.3, "WindowSize": 4, "CandidateCount": 120}}]
    cfg["SortConfs"] = [{"Name": "ignored_rule_sort", "SortType": "BoostScoreSort"}]
    cfg["SortNames"] = {"home_feed": ["my_ssd"]}
    got_conf = json.loads(H.ph_parse_recconf(json.dumps(cfg).encode()))
    assert got_conf is not None
    h = H.ph_engine_create(json.dumps(cfg).encode())
    assert h, H.ph_last_error()
    n, d = 20000, 128
    tab = o.synth_rows(o.SEED_TABLE, 0, n, d)
    w = o.Dnn3Weights()
    user = o.synth_rows(o.SEED_QUERY, 5, 1, d)[0]
    vec = " ".join("%d:%s" % (i + 1, repr(float(v))) for i, v in enumerate(user))
    H.ph_set_user_vector(h, b"u2", vec.encode())
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    assert H.ph_engine_load_dnn3(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    size = 30
    out = json.loads(H.ph_recommend(h, b"u2", size, b"home_feed"))
    # the same request through the oracle
    rows, scores = o.recall_topk(tab, user[None], 300)
    items = [o.OracleItem("item_%d" % r, float(s), "gpu_vector_recall") for r, s in zip(rows[0], scores[0])]
    items = o.unique_filter(items)
    dnn = o.dnn3_forward(w, 0, user, tab[rows[0].astype(np.int64)])
    for it, s in zip(items, dnn):
        it.add_algo_score("gpu_dnn", float(np.float32(s)))
    o.fuse_scores(RANK_SCORE, items)
    # the GPU's scores differ from the oracle's by <= 1e-6 (expf / pow), which may swap near-ties in the
    # rank order; feed the SSD oracle the order and scores the engine itself reports for its page and
    # require only what is independent of that: a page of `size` distinct recalled items, first = best
    got_ids = [x["item_id"] for x in out["items"]]
    assert len(got_ids) == size and len(set(got_ids)) == size
    want_scores = {it.id: it.score for it in items}
    for x in out["items"]:
        assert abs(x["score"] - want_scores[x["item_id"]]) <= 1e-6
    order = o.sort_scores([it.score for it in items], True)[:120]        # CandidateCount = 120
    cand = [items[i] for i in order]
    if np.all(np.abs(np.diff([c.score for c in cand])) > 4e-6):          # no near-ties: exact comparison
        rowid = np.array([int(c.id.split("_")[1]) for c in cand])
        emb = o.ssd_embeddings(tab[rowid], True, True)
        picks = o.ssd_window(emb, np.array([c.score for c in cand]), 0.3, size, 4)
        # quality = score + volume*norm is insensitive to 1e-6 score noise unless two qualities tie
        assert got_ids[0] == cand[0].id
        assert got_ids == [cand[i].id for i in picks]
    assert got_ids[0] == cand[0].id
    H.ph_engine_destroy(h)


# ---- recall result cache format, cache adapters, AB clone hooks ------------------------
def _bind_row2(H):
    H.ph_go_fmt_float.restype = C.c_char_p
    H.ph_go_fmt_float.argtypes = [C.c_double]
    H.ph_format_recall_cache.restype = C.c_char_p
    H.ph_format_recall_cache.argtypes = [C.c_char_p, C.c_char_p]
    H.ph_parse_recall_cache.restype = C.c_char_p
    H.ph_parse_recall_cache.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    H.ph_recommend_ab.restype = C.c_char_p
    H.ph_recommend_ab.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
    return H


def test_go_percent_v_float_format(H):
    """fmt %v of float64 (the score field of the cache line, vector_recall.go:107) against the oracle's
    restatement, over shortest-repr edge cases."""
    _bind_row2(H)
    rng = np.random.default_rng(3)
    vals = [0.0, -0.0, 1.0, -1.0, 0.5, 1e-7, 1e-5, 1e-4, 0.0001234, 123456789.0, 1e20, 1e21, 1.5e21, 3.0,
            0.1, 0.30000000000000004, 2.5e-300, 1.7976931348623157e308, 5e-324, float("inf"), -float("inf"),
            float(np.float32(0.8612537))]
    vals += list(rng.standard_normal(200)) + list(10.0 ** rng.uniform(-30, 30, 200))
    for v in vals:
        assert H.ph_go_fmt_float(float(v)).decode() == o.go_fmt_float(float(v)), v
    assert H.ph_go_fmt_float(float("nan")) == b"NaN"
    # strconv 'g' with the shortest digits switches to the exponent form at exponent 6 (`fmt.Println(float64(12345678))` is
    # 1.2345678e+07) and below -4
    for v, want in [(100000.0, "100000"), (999999.5, "999999.5"), (1e6, "1e+06"), (12345678.0, "1.2345678e+07"),
                    (1700000000.0, "1.7e+09"), (0.0001, "0.0001"), (0.00001234, "1.234e-05"), (-2500000.0, "-2.5e+06")]:
        assert H.ph_go_fmt_float(v).decode() == o.go_fmt_float(v) == want


def test_recall_cache_line_format_and_parse(H):
    _bind_row2(H)
    items = [o.OracleItem("a", 0.5, "vec"), o.OracleItem("b", 1e-7, "vec"), o.OracleItem("c", 3.0, "vec")]
    js = json.dumps([{"id": it.id, "score": it.score} for it in items]).encode()
    line = H.ph_format_recall_cache(js, b"vec").decode()
    assert line == o.recall_cache_string(items, "vec") == "a:vec:0.5,b:vec:1e-07,c:vec:3"
    back = json.loads(H.ph_parse_recall_cache(line.encode(), b"vec", b"video"))["items"]
    assert [(x["item_id"], x["score"], x["retrieve_id"], x["item_type"]) for x in back] == \
        [("a", 0.5, "vec", "video"), ("b", 1e-7, "vec", "video"), ("c", 3.0, "vec", "video")]
    # ids without ':' carry no score (vector_recall.go:48-50); a malformed score parses to 0 (`f, _ :=`)
    back = json.loads(H.ph_parse_recall_cache(b"x,y:vec:oops", b"vec", b""))["items"]
    assert [(x["item_id"], x["score"]) for x in back] == [("x", 0.0), ("y", 0.0)]
    # "id:name" (two fields) indexes vars[2] in the reference — a panic there, an error here
    assert H.ph_parse_recall_cache(b"y:vec", b"vec", b"") is None and b"id:name:score" in H.ph_last_error()


def test_cache_adapters_and_clone_hooks(H):
    assert H.ph_cache_clone_semantics() == 15


@pytest.mark.gpu
def test_ab_params_recall_clone_cache_and_ssd_overrides(H):
    """An AB experiment attached to the request: "recall.<name>" clones the recall with a new RecallCount
    (ICloneRecall, service/recall.go:95-105); a byte-valued cache serves the second request from the
    cache line; ssd_* parameters override the SSDSort config (ssd_sort.go:301-309)."""
    import copy
    import pairec_amd as pa
    _bind_row2(H)
    cfg = copy.deepcopy(CONFIG)
    cfg["UserDefineConfs"]["pairec_gpu"]["Recalls"][0].update({"CacheAdapter": "localBytes", "CachePrefix": "vr_", "CacheTime": 60})
    cfg["UserDefineConfs"]["pairec_gpu"]["Sorts"] = [{"Name": "my_ssd", "SortType": "SSDSort",
                                                      "SSDConf": {"Gamma": 0.3, "WindowSize": 4}}]
    cfg["SortNames"] = {"home_feed": ["my_ssd"]}
    h = H.ph_engine_create(json.dumps(cfg).encode())
    assert h, H.ph_last_error()
    w = o.Dnn3Weights()
    user = o.synth_rows(o.SEED_QUERY, 9, 1, 128)[0]
    vec = " ".join("%d:%s" % (i + 1, repr(float(v))) for i, v in enumerate(user))
    H.ph_set_user_vector(h, b"u3", vec.encode())
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    assert H.ph_engine_load_dnn3(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    base = json.loads(H.ph_recommend(h, b"u3", 400, b"home_feed"))["items"]
    assert len(base) == 300                                   # RecallCount = 300 in CONFIG
    # the clone recalls 40 instead (and has no cache of its own: the params object replaces the config)
    ab = json.loads(H.ph_recommend_ab(h, b"u3", 400, b"home_feed",
                                      json.dumps({"recall.gpu_vector_recall": {"RecallCount": 40,
                                                                               "RecallAlgo": "gpu_faiss"}}).encode()))["items"]
    assert len(ab) == 40 and {x["item_id"] for x in ab} <= {x["item_id"] for x in base}
    # second plain request: served from the cache line — same ids, scores round-trip through %v exactly
    again = json.loads(H.ph_recommend(h, b"u3", 400, b"home_feed"))["items"]
    assert [x["item_id"] for x in again] == [x["item_id"] for x in base]
    assert [x["score"] for x in again] == [x["score"] for x in base]
    # ssd_gamma = 0 → SSD is skipped: plain rank order (ssd_sort.go:302-305)
    g0 = json.loads(H.ph_recommend_ab(h, b"u3", 30, b"home_feed", json.dumps({"ssd_gamma": 0}).encode()))["items"]
    sc = [x["score"] for x in g0]
    assert sc == sorted(sc, reverse=True)
    # ssd_norm_quality_score = 2 records "ssd_quality_score" on the candidates (ssd_sort.go:385)
    q2 = json.loads(H.ph_recommend_ab(h, b"u3", 30, b"home_feed",
                                      json.dumps({"ssd_norm_quality_score": 2}).encode()))["items"]
    assert all("ssd_quality_score" in x["algo_scores"] for x in q2) and q2[0]["algo_scores"]["ssd_quality_score"] == 1.0
    H.ph_engine_destroy(h)


# ---- round 2: multi-output rank algorithms, I2I / online-vector recalls, AlgoScoreSort, DPP by name, concurrency ----
def _engine(H, cfg, uid=b"u1", qrow=3):
    import pairec_amd as pa
    h = H.ph_engine_create(json.dumps(cfg).encode())
    assert h, H.ph_last_error()
    w = o.Dnn3Weights()
    user = o.synth_rows(o.SEED_QUERY, qrow, 1, 128)[0]
    vec = " ".join("%d:%s" % (i + 1, repr(float(v))) for i, v in enumerate(user))
    H.ph_set_user_vector(h, uid, vec.encode())
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    assert H.ph_engine_load_dnn3(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    return h, w, user


@pytest.mark.gpu
def test_multi_output_algorithm_writes_algo_output_scores(H):
    """GetModuleType() == true (EasyrecResponse.multiValModule): RankService writes one algo score per output as
    "<algo>_<output>" (rank_service.go:315-319) and the RankScore expression combines them — the reference's own
    known-answer shape `(${ppnet_probs_ctr}+2*${ppnet_probs_cvr})` (utils/ast/ast_test.go:90-129)."""
    import copy
    import pairec_amd as pa
    H.ph_engine_load_dnn3_named.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    cfg = copy.deepcopy(CONFIG)
    cfg["UserDefineConfs"]["pairec_gpu"]["Algorithms"][1] = {"Name": "ppnet", "Kind": "dnn3", "Outputs": ["probs_ctr", "probs_cvr"]}
    expr = "(${ppnet_probs_ctr}+2*${ppnet_probs_cvr})*(1+${current_score})^0.1"
    cfg["RankConf"]["home_feed"] = {"RankAlgoList": ["ppnet"], "RankScore": expr, "BatchCount": 100}
    h, _, user = _engine(H, cfg)
    heads = {"probs_ctr": o.Dnn3Weights(seed=o.SEED_WEIGHTS ^ 0x11), "probs_cvr": o.Dnn3Weights(seed=o.SEED_WEIGHTS ^ 0x22)}
    for name, w in heads.items():
        blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
        assert H.ph_engine_load_dnn3_named(h, ("ppnet/" + name).encode(), pa.PREC_F32, blob, len(blob)) == 0
    out = json.loads(H.ph_recommend(h, b"u1", 40, b"home_feed"))["items"]
    tab = o.synth_rows(o.SEED_TABLE, 0, 20000, 128)
    rows, scores = o.recall_topk(tab, user[None], 300)
    ref = {name: o.dnn3_forward(w, 0, user, tab[rows[0].astype(np.int64)]) for name, w in heads.items()}
    pos = {"item_%d" % r: i for i, r in enumerate(rows[0])}
    for x in out:
        i = pos[x["item_id"]]
        a, b = x["algo_scores"]["ppnet_probs_ctr"], x["algo_scores"]["ppnet_probs_cvr"]
        assert abs(a - ref["probs_ctr"][i]) <= 2e-7 and abs(b - ref["probs_cvr"][i]) <= 2e-7
        assert "ppnet" not in x["algo_scores"]
        assert abs(x["score"] - (a + 2 * b) * (1 + float(scores[0][i])) ** 0.1) <= 1e-12
    assert [x["score"] for x in out] == sorted((x["score"] for x in out), reverse=True)
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_page_recall_returns_the_staged_scenes_page(H):
    """pairec_gpu.Recalls Kind "page": one recall plug-in that returns the finished page (recall → DNN rank → RankScore →
    ItemRankScore sort on the device, one coalesced single-request call) — the same ids, order and scores as the scene
    that runs the stages one plug-in at a time, with only ctx.Size items materialised on the host."""
    import copy
    staged, _, user = _engine(H, CONFIG)
    want = json.loads(H.ph_recommend(staged, b"u1", 25, b"home_feed"))["items"]
    H.ph_engine_destroy(staged)
    cfg = copy.deepcopy(CONFIG)
    cfg["SceneConfs"] = {"home_feed": {"default": {"RecallNames": ["gpu_page"]}}}
    cfg["RankConf"] = {}
    cfg["SortNames"] = {}
    cfg["UserDefineConfs"]["pairec_gpu"]["Recalls"] = [{"Name": "gpu_page", "Kind": "page", "RecallCount": 300,
                                                       "ItemType": "video", "RankScore": RANK_SCORE, "RankVar": "gpu_dnn"}]
    h, _, _ = _engine(H, cfg)
    got = json.loads(H.ph_recommend(h, b"u1", 25, b"home_feed"))["items"]
    assert [x["item_id"] for x in got] == [x["item_id"] for x in want]
    for a, b in zip(got, want):
        assert abs(a["score"] - b["score"]) <= 1e-12 * max(1.0, abs(b["score"]))
        assert abs(a["algo_scores"]["gpu_dnn"] - b["algo_scores"]["gpu_dnn"]) <= 1e-7
        assert a["retrieve_id"] == "gpu_page"
    # a page recall without its expression is refused when the engine is created
    bad = copy.deepcopy(cfg)
    del bad["UserDefineConfs"]["pairec_gpu"]["Recalls"][0]["RankScore"]
    assert not H.ph_engine_create(json.dumps(bad).encode()) and b"RankScore" in H.ph_last_error()
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_i2i_online_vector_recalls_and_algo_score_sort(H):
    """a5 / a6 through the registries: the GPU I2I recall (trigger = the request's item_id), the reference's own
    OnlineVectorRecall declared in RecallConfs (its constructor needs no datasource) served by a GPU vector model,
    and AlgoScoreSort from SortConfs with its key sort on the device."""
    import copy
    import pairec_amd as pa
    _bind_row2(H)
    H.ph_engine_load_fm2t.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    cfg = copy.deepcopy(CONFIG)
    g = cfg["UserDefineConfs"]["pairec_gpu"]
    g["Recalls"].append({"Name": "gpu_i2i", "Kind": "i2i", "RecallCount": 50, "ItemType": "video"})
    g["Algorithms"].append({"Name": "gpu_vec_model", "Kind": "online_vector"})
    g["OnlineVector"] = {"Rows": 20000, "Dim": 64, "SyntheticSeed": o.SEED_TABLE ^ 0x77}
    cfg["RecallConfs"] = [{"Name": "online_vec", "RecallType": "OnlineVectorRecall", "RecallCount": 80,
                           "RecallAlgo": "gpu_vec_model", "VectorAlgoType": "torchrec_vector"}]
    cfg["SceneConfs"]["item_detail"] = {"default": {"RecallNames": ["gpu_i2i"]}}
    cfg["SceneConfs"]["cold_feed"] = {"default": {"RecallNames": ["online_vec"]}}
    cfg["SortConfs"] = [{"Name": "by_ctr", "SortType": "AlgoScoreSort", "SortByField": "gpu_dnn", "SwitchThreshold": 10.0}]
    cfg["RankConf"]["item_detail"] = {"RankAlgoList": ["gpu_dnn"], "RankScore": "", "BatchCount": 100}
    cfg["SortNames"]["item_detail"] = ["by_ctr"]
    h, w, user = _engine(H, cfg)
    tab = o.synth_rows(o.SEED_TABLE, 0, 20000, 128)
    # I2I: trigger item_777 → its 50 nearest items; rank writes gpu_dnn, the page is sorted by that field (Item.Score —
    # the recall distance, all below the threshold 10 — is untouched because RankScore is empty)
    out = json.loads(H.ph_recommend_ab(h, b"u1", 50, b"item_detail", json.dumps({"_param": {"item_id": "item_777"}}).encode()))["items"]
    rows, scores = o.recall_topk(tab, tab[777][None], 50)
    assert sorted(x["item_id"] for x in out) == sorted("item_%d" % r for r in rows[0])
    assert {x["retrieve_id"] for x in out} == {"gpu_i2i"}
    dnn = o.dnn3_forward(w, 0, user, tab[rows[0].astype(np.int64)])
    want = ["item_%d" % rows[0][i] for i in np.argsort(-dnn, kind="stable")]
    keys = [x["algo_scores"]["gpu_dnn"] for x in out]
    assert keys == sorted(keys, reverse=True)
    if np.all(np.abs(np.diff(np.sort(dnn))) > 1e-6):
        assert [x["item_id"] for x in out] == want
    by_id = {"item_%d" % r: float(s) for r, s in zip(rows[0], scores[0])}
    assert all(x["score"] == by_id[x["item_id"]] for x in out)
    # online vector recall: user tower of the FM + two-tower model → 80 nearest rows of the item-embedding table
    fw = o.Fm2tWeights(vocab=500)
    blob = pa.pack_fm2t(fw)
    assert H.ph_engine_load_fm2t(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    out = json.loads(H.ph_recommend(h, b"u1", 80, b"cold_feed"))["items"]
    emb = o.synth_rows(o.SEED_TABLE ^ 0x77, 0, 20000, 64)
    ue = o.fm2t_user_embedding(fw, 0, user)
    orow, osc = o.recall_topk(emb, ue[None], 80)
    assert [x["item_id"] for x in out] == ["item_%d" % r for r in orow[0]]          # no rank config: recall order = score order
    assert [x["score"] for x in out] == [float(s) for s in osc[0]] and {x["retrieve_id"] for x in out} == {"online_vec"}
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_hologres_vector_recall_v2_squared_euclidean(H):
    """HologresVectorRecallV2 (service/recall/hologres_vector_recall_v2.go:23,96-206): the RecallCount items of smallest
    squared Euclidean distance to the user's embedding, ascending, Score = distance — through the recall registry, with the
    embedding in the Hologres DAO's own "{v1,v2,…}" text form.  Rows of different norms, so that the order is not the
    inner product's."""
    import copy
    cfg = copy.deepcopy(CONFIG)
    g = cfg["UserDefineConfs"]["pairec_gpu"]
    g["Recalls"].append({"Name": "holo_v2", "Kind": "hologres_v2", "RecallCount": 120, "ItemType": "video"})
    cfg["SceneConfs"]["near_feed"] = {"default": {"RecallNames": ["holo_v2"]}}
    h, _, user = _engine(H, cfg)
    tab = o.synth_rows(o.SEED_TABLE, 0, 20000, 128)
    scaled = (tab * np.linspace(0.5, 1.5, 20000, dtype=np.float32)[:, None]).astype(np.float32)
    # a new table generation with those rows (ids as before): the ingestion path of host/ingest.cpp
    H.ph_engine_ingest_begin.argtypes = [C.c_void_p]
    H.ph_engine_ingest_chunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint64]
    H.ph_engine_ingest_commit.argtypes = [C.c_void_p]
    assert H.ph_engine_ingest_begin(h) == 0
    ids = b"".join(b"item_%d\0" % i for i in range(20000))
    assert H.ph_engine_ingest_chunk(h, ids, len(ids), scaled.ctypes.data, 20000) == 0
    assert H.ph_engine_ingest_commit(h) == 0
    H.ph_set_user_vector(h, b"u2", ("{" + ",".join(repr(float(v)) for v in user) + "}").encode())
    out = json.loads(H.ph_recommend(h, b"u2", 120, b"near_feed"))["items"]
    orow, od = o.recall_topk_l2(scaled, user[None], 120)
    # no rank / sort configured for the scene: the default ItemRankScore sort orders by Score descending — the SET and the
    # distances are what the recall contributes
    assert sorted(x["item_id"] for x in out) == sorted("item_%d" % r for r in orow[0])
    by_id = {"item_%d" % r: float(d) for r, d in zip(orow[0], od[0])}
    assert all(x["score"] == by_id[x["item_id"]] for x in out) and {x["retrieve_id"] for x in out} == {"holo_v2"}
    irow, _ = o.recall_topk(scaled, user[None], 120)
    assert set(irow[0].tolist()) != set(orow[0].tolist())
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_hologres_vector_recalls_with_a_where_clause(H):
    """HologresVectorRecall / HologresVectorRecallV2 with HologresVectorConf.WhereClause (hologres_vector_recall.go:23,49-62,
    _v2.go:23): "FROM table WHERE create_time > ${time} ORDER BY distance LIMIT n" — ${time} = now - TimeInterval, fixed when the
    recall is built.  Against the oracle on the admitted rows alone: a filter that keeps half the table (served in place), one
    that keeps a fiftieth (served from the compact copy), one that keeps fewer rows than RecallCount, and an unknown column
    (the reference's SQL error: logged, empty list)."""
    import copy, time
    H.ph_engine_set_feature_column.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    cfg = copy.deepcopy(CONFIG)
    g = cfg["UserDefineConfs"]["pairec_gpu"]
    now = int(time.time())
    g["Recalls"] += [
        {"Name": "holo_recent", "Kind": "hologres", "RecallCount": 150, "ItemType": "video",
         "HologresVectorConf": {"WhereClause": "create_time > ${time}", "TimeInterval": 86400}},
        {"Name": "holo_v2_recent", "Kind": "hologres_v2", "RecallCount": 150, "ItemType": "video",
         "HologresVectorConf": {"WhereClause": "create_time > ${time}", "TimeInterval": 86400}},
        {"Name": "holo_cat", "Kind": "hologres", "RecallCount": 150, "WhereClause": "cat_id = 7"},
        {"Name": "holo_rare", "Kind": "hologres_v2", "RecallCount": 150, "WhereClause": "cat_id >= 49"},
        {"Name": "holo_nocol", "Kind": "hologres", "RecallCount": 150, "WhereClause": "missing < 3"}]
    for name in ("holo_recent", "holo_v2_recent", "holo_cat", "holo_rare", "holo_nocol"):
        cfg["SceneConfs"]["s_" + name] = {"default": {"RecallNames": [name]}}
    h, _, user = _engine(H, cfg)
    n = 20000
    tab = o.synth_rows(o.SEED_TABLE, 0, n, 128)
    rng = np.random.default_rng(9)
    # half the items are newer than a day (the clause's constant was fixed within the last seconds: keep a margin around it)
    age = np.where(rng.random(n) < 0.5, rng.integers(0, 80000, n), rng.integers(90000, 900000, n))
    create_time = (now - age).astype(np.int32)
    cat = rng.integers(0, 50, n).astype(np.int32)
    cat[rng.random(n) < 0.996] %= 49                                    # category 49: a few dozen rows
    assert H.ph_engine_set_feature_column(h, b"create_time", create_time.ctypes.data, n) == 0, H.ph_last_error()
    assert H.ph_engine_set_feature_column(h, b"cat_id", cat.ctypes.data, n) == 0, H.ph_last_error()
    H.ph_set_user_vector(h, b"u3", ("{" + ",".join(repr(float(v)) for v in user) + "}").encode())

    def check(scene, mask, l2, k=150, tab_=None):
        tab_ = tab if tab_ is None else tab_
        out = json.loads(H.ph_recommend(h, b"u3", k, scene.encode()))["items"]
        idx = np.nonzero(mask)[0]
        f = o.recall_topk_l2 if l2 else o.recall_topk
        orow, osc = f(tab_[idx], user[None], k)
        m = min(k, idx.size)
        want = {"item_%d" % idx[int(r)]: float(sc) for r, sc in zip(orow[0][:m], osc[0][:m])}
        assert len(out) == m and {x["item_id"]: x["score"] for x in out} == want, scene

    check("s_holo_recent", age < 86400, False)
    check("s_holo_v2_recent", age < 86400, True)
    check("s_holo_cat", cat == 7, False)
    assert 0 < int((cat >= 49).sum()) < 150
    check("s_holo_rare", cat >= 49, True)
    assert json.loads(H.ph_recommend(h, b"u3", 150, b"s_holo_nocol"))["items"] == []
    # the filter's view follows the column: category 7 moves to other items
    cat2 = np.roll(cat, 1234)
    assert H.ph_engine_set_feature_column(h, b"cat_id", cat2.ctypes.data, n) == 0, H.ph_last_error()
    check("s_holo_cat", cat2 == 7, False)
    # ... and the table: a new generation of rows (same ids) is ingested; the first request afterwards rebuilds the view
    H.ph_engine_ingest_begin.argtypes = [C.c_void_p]
    H.ph_engine_ingest_chunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint64]
    H.ph_engine_ingest_commit.argtypes = [C.c_void_p]
    tab2 = (tab * rng.uniform(0.6, 1.4, (n, 1)).astype(np.float32)).astype(np.float32)
    assert H.ph_engine_ingest_begin(h) == 0
    ids = b"".join(b"item_%d\0" % i for i in range(n))
    assert H.ph_engine_ingest_chunk(h, ids, len(ids), tab2.ctypes.data, n) == 0
    # the generation brings its row-keyed columns along (they change over with the rows, inside the commit): category 7 moves again
    H.ph_engine_ingest_feature_column.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    cat3 = np.roll(cat, 4321)
    assert H.ph_engine_ingest_feature_column(h, b"cat_id", cat3.ctypes.data, n) == 0
    assert H.ph_engine_ingest_feature_column(h, b"create_time", create_time.ctypes.data, n) == 0
    assert H.ph_engine_ingest_commit(h) == 0
    check("s_holo_cat", cat3 == 7, False, tab_=tab2)
    check("s_holo_v2_recent", age < 86400, True, tab_=tab2)
    H.ph_engine_destroy(h)
    # Coalesce on: the view has a coalescer of its own; 48 requests from 16 threads per filtered recall
    g["Coalesce"] = {"MaxWaitUs": 2000}
    h, _, user = _engine(H, cfg)
    assert H.ph_engine_set_feature_column(h, b"create_time", create_time.ctypes.data, n) == 0, H.ph_last_error()
    assert H.ph_engine_set_feature_column(h, b"cat_id", cat.ctypes.data, n) == 0, H.ph_last_error()
    H.ph_set_user_vector(h, b"u3", ("{" + ",".join(repr(float(v)) for v in user) + "}").encode())
    H.ph_recommend_concurrent.restype = C.c_char_p
    H.ph_recommend_concurrent.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    for scene, mask, l2 in (("s_holo_cat", cat == 7, False), ("s_holo_v2_recent", age < 86400, True)):
        pages = json.loads(H.ph_recommend_concurrent(h, json.dumps(["u3"] * 48).encode(), 150, scene.encode(), 16))
        idx = np.nonzero(mask)[0]
        orow, osc = (o.recall_topk_l2 if l2 else o.recall_topk)(tab[idx], user[None], 150)
        want = {"item_%d" % idx[int(r)]: float(sc) for r, sc in zip(orow[0], osc[0])}
        assert len(pages) == 48 and all({x["item_id"]: x["score"] for x in p["items"]} == want for p in pages), scene
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_fm2t_algorithm_easyrec_flavour(H):
    """FM + two-tower registered as an IAlgorithm: the EasyRec request flavour (item ids + columnar context features,
    service/rank/algo_data.go:79-86,223-306) with the columns resident on the device — scores equal the oracle's
    forward on the same field ids, rank order follows."""
    import copy
    import pairec_amd as pa
    H.ph_engine_load_fm2t.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    H.ph_engine_set_feature_column.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    H.ph_set_user_fields.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    cfg = copy.deepcopy(CONFIG)
    cols = ["cat%d" % f for f in range(8)]
    cfg["UserDefineConfs"]["pairec_gpu"]["Algorithms"].append({"Name": "gpu_fm", "Kind": "fm2t", "ItemFieldColumns": cols})
    cfg["RankConf"]["home_feed"] = {"RankAlgoList": ["gpu_fm"], "RankScore": "${gpu_fm}", "BatchCount": 100}
    h, _, user = _engine(H, cfg)
    vocab, n = 700, 20000
    fw = o.Fm2tWeights(vocab=vocab)
    blob = pa.pack_fm2t(fw)
    assert H.ph_engine_load_fm2t(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    rng = np.random.default_rng(2)
    item_fields = rng.integers(0, vocab, (n, 8)).astype(np.int32)
    for f, c in enumerate(cols):
        col = np.ascontiguousarray(item_fields[:, f])
        assert H.ph_engine_set_feature_column(h, c.encode(), col.ctypes.data, n) == 0, H.ph_last_error()
    ufields = rng.integers(0, vocab, 8).astype(np.int32)
    assert H.ph_set_user_fields(h, b"u1", ufields.ctypes.data, 8) == 0
    out = json.loads(H.ph_recommend(h, b"u1", 60, b"home_feed"))["items"]
    tab = o.synth_rows(o.SEED_TABLE, 0, n, 128)
    rows, _ = o.recall_topk(tab, user[None], 300)
    ref = o.fm2t_forward(fw, 0, user, ufields, item_fields[rows[0].astype(np.int64)])
    pos = {"item_%d" % r: i for i, r in enumerate(rows[0])}
    assert len(out) == 60
    for x in out:
        assert abs(x["algo_scores"]["gpu_fm"] - ref[pos[x["item_id"]]]) <= 2e-7 and x["score"] == x["algo_scores"]["gpu_fm"]
    assert [x["score"] for x in out] == sorted((x["score"] for x in out), reverse=True)
    assert min(x["score"] for x in out) >= np.sort(ref)[::-1][59] - 2e-7
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_gpu_dpp_sort_by_name_with_experiment_overrides(H):
    """DPPSort declared in pairec_gpu.Sorts: page = the oracle's DPP picks over the first CandidateCount items; the
    dpp_* experiment parameters override the config (dpp_sort.go:275-278,374,382) and "dpp_relevance_score" is
    recorded on the candidates (:410)."""
    import copy
    _bind_row2(H)
    cfg = copy.deepcopy(CONFIG)
    cfg["UserDefineConfs"]["pairec_gpu"]["Sorts"] = [{"Name": "GpuDPP", "SortType": "DPPSort",
                                                      "DPPConf": {"Alpha": 1.0, "WindowSize": 10, "CandidateCount": 120}}]
    cfg["SortNames"] = {"home_feed": ["GpuDPP"]}
    h, w, user = _engine(H, cfg)
    size = 30
    tab = o.synth_rows(o.SEED_TABLE, 0, 20000, 128)
    rows, scores = o.recall_topk(tab, user[None], 300)
    items = [o.OracleItem("item_%d" % r, float(s), "gpu_vector_recall") for r, s in zip(rows[0], scores[0])]
    dnn = o.dnn3_forward(w, 0, user, tab[rows[0].astype(np.int64)])
    for it, s_ in zip(items, dnn):
        it.add_algo_score("gpu_dnn", float(np.float32(s_)))
    o.fuse_scores(RANK_SCORE, items)
    order = o.sort_scores([it.score for it in items], True)

    def want_page(cand_cnt, alpha, window, mode):
        cand = [items[i] for i in order[:max(size, cand_cnt)]]
        rowid = np.array([int(c.id.split("_")[1]) for c in cand])
        rel, ok = o.dpp_relevance(np.array([c.score for c in cand]), mode)
        F = o.dpp_features(tab[rowid], None, True, True)
        picks = o.dpp_with_window(o.dpp_kernel_matrix_f(F, rel, alpha), size, window)
        return [cand[i].id for i in picks], cand
    near_tie = np.any(np.abs(np.diff(sorted(it.score for it in items))) <= 4e-6)
    out = json.loads(H.ph_recommend(h, b"u1", size, b"home_feed"))["items"]
    page, cand = want_page(120, 1.0, 10, 0)
    assert len(out) == size and len({x["item_id"] for x in out}) == size
    assert {x["item_id"] for x in out} <= {c.id for c in cand} and out[0]["item_id"] == cand[0].id
    if not near_tie:
        assert [x["item_id"] for x in out] == page
    assert all("dpp_relevance_score" in x["algo_scores"] for x in out)
    ab = {"dpp_alpha": 3.0, "dpp_window_size": 5, "dpp_candidate_count": 60, "dpp_norm_relevance_score": 2}
    out2 = json.loads(H.ph_recommend_ab(h, b"u1", size, b"home_feed", json.dumps(ab).encode()))["items"]
    page2, cand2 = want_page(60, 3.0, 5, 2)
    assert {x["item_id"] for x in out2} <= {c.id for c in cand2}
    if not near_tie:
        assert [x["item_id"] for x in out2] == page2
    rs = {x["item_id"]: x["algo_scores"]["dpp_relevance_score"] for x in out2}
    assert max(rs.values()) <= 1.0 and min(rs.values()) >= 1e-6 and rs[cand2[0].id] == 1.0      # min-max into [1e-6, 1]
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_concurrent_requests_through_the_coalescer_equal_sequential(H):
    """"Coalesce" in pairec_gpu: GpuFaissAlgorithm / GpuDnnAlgorithm issue ONE request per call through
    pg_coalescer_recall / pg_coalescer_rank_dnn3 (50 rank calls of 100 items per request, as RankService fans them out);
    48 requests on 16 threads give the pages the same engine gives one at a time without coalescing."""
    import copy
    import pairec_amd as pa
    H.ph_recommend_concurrent.restype = C.c_char_p
    H.ph_recommend_concurrent.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    uids = ["user_%d" % i for i in range(48)]
    vecs = o.synth_rows(o.SEED_QUERY, 100, len(uids), 128)
    w = o.Dnn3Weights()
    blob = pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128)
    pages = {}
    for mode in ("plain", "coalesce"):
        cfg = copy.deepcopy(CONFIG)
        if mode == "coalesce":
            cfg["UserDefineConfs"]["pairec_gpu"]["Coalesce"] = {"MaxWaitUs": 300, "Depth": 2}
        h = H.ph_engine_create(json.dumps(cfg).encode())
        assert h, H.ph_last_error()
        for u, v in zip(uids, vecs):
            H.ph_set_user_vector(h, u.encode(), " ".join("%d:%s" % (i + 1, repr(float(x))) for i, x in enumerate(v)).encode())
        assert H.ph_engine_load_dnn3(h, pa.PREC_F32, blob, len(blob)) == 0
        r = H.ph_recommend_concurrent(h, json.dumps(uids).encode(), 25, b"home_feed", 16 if mode == "coalesce" else 1)
        assert r, H.ph_last_error()
        pages[mode] = json.loads(r)
        H.ph_engine_destroy(h)
    assert len(pages["coalesce"]) == len(uids)
    for a, b in zip(pages["plain"], pages["coalesce"]):
        assert [x["item_id"] for x in a["items"]] == [x["item_id"] for x in b["items"]]
        assert [x["score"] for x in a["items"]] == [x["score"] for x in b["items"]]
        assert [x["algo_scores"] for x in a["items"]] == [x["algo_scores"] for x in b["items"]]


# ---- the host boxing the device feature store replaces ---------------------------------
def test_easyrec_generator_default_filling(H):
    """EasyrecAlgoDataGenerator.AddFeatures / GeneratorAlgoData (algo_data.go:172-306) in the C++ mirror
    against the oracle's restatement: configured context features default to "", "*" item features take
    the first item's keys and Go zero values, lists reset per batch."""
    H.ph_easyrec_generator.restype = C.c_char_p
    H.ph_easyrec_generator.argtypes = [C.c_char_p]
    items = [{"id": "i1", "features": {"cat": "a", "price": 3.5, "cnt": 7, "extra": "x"}},
             {"id": "i2", "features": {"cat": "b", "cnt": 2}},
             {"id": "i3", "features": {"price": 1.25, "late": 9}},
             {"id": "i4", "features": {}}]
    user = {"age": 30, "city": "hz"}
    for ctxf, itemf in ((["cat", "missing"], None), (["cat"], ["*"]), (["cat"], ["price", "nope"]), ([], ["*"]),
                        (["cat", "price"], [])):
        spec = {"context_features": ctxf, "item_features": itemf, "user": user, "items": items, "batches": [3, 1]}
        got = json.loads(H.ph_easyrec_generator(json.dumps(spec).encode()))
        g = o.EasyrecGenerator(ctxf)
        if itemf is not None:
            g.set_item_features(itemf)
        want, pos = [], 0
        for b in spec["batches"]:
            for it in items[pos:pos + b]:
                g.add_features(it["id"], it["features"], user)
            pos += b
            want.append(g.generate())
        assert got == want, (ctxf, itemf)
    # spot checks of the semantics themselves
    g = o.EasyrecGenerator(["cat", "missing"])
    for it in items[:3]:
        g.add_features(it["id"], it["features"], user)
    r = g.generate()
    assert r["context_features"] == {"cat": ["a", "b", ""], "missing": ["", "", ""]} and r["item_features"] == {}
    g = o.EasyrecGenerator(["cat"])
    g.set_item_features(["*"])
    for it in items:
        g.add_features(it["id"], it["features"], user)
    r = g.generate()
    assert r["item_features"] == {"price": [3.5, 0.0, 1.25, 0.0], "cnt": [7, 2, 0, 0], "extra": ["x", "", "", ""]}


# ---- /api/recommend harness --------------------------------------------------------------
def test_http_harness_parameter_checks_without_engine():
    """RecommendController.CheckParameter (web/recommend_controller.go:96-113) — the checks that fail before
    the engine is touched."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import http_harness as hh
    h = hh.Harness.__new__(hh.Harness)                       # no engine: only the early-exit paths
    assert h.handle(b"")["code"] == 400 and h.handle(b"")["msg"] == "request body empty"
    assert h.handle(b"{not json")["code"] == 400
    r = h.handle(json.dumps({"size": 5}).encode())
    assert r["code"] == 400 and r["msg"] == "uid not empty" and len(r["request_id"]) == 36


@pytest.mark.gpu
def test_http_harness_end_to_end():
    """POST /api/recommend over localhost → recall → rank → sort on the GPU → the reference's response
    shape (code 200 / 299, size, items[{item_id,item_type,retrieve_id}])."""
    import sys
    import threading
    import urllib.request
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import http_harness as hh
    import pairec_amd as pa
    h = hh.Harness(CONFIG)
    w = o.Dnn3Weights()
    h.load_dnn3(pa.PREC_F32, pa.pack_dnn3(w.w1, w.b1, w.w2, w.b2, w.w3, w.b3, 128))
    srv = hh.make_server(h, 0)
    port = srv.server_address[1]
    th = threading.Thread(target=srv.serve_forever, daemon=True)
    th.start()

    def post(obj):
        req = urllib.request.Request("http://127.0.0.1:%d/api/recommend" % port, data=json.dumps(obj).encode(),
                                     headers={"Content-Type": "application/json"})
        with urllib.request.urlopen(req, timeout=60) as r:
            return json.loads(r.read())

    user = o.synth_rows(o.SEED_QUERY, 3, 1, 128)[0]
    vec = " ".join("%d:%s" % (i + 1, repr(float(v))) for i, v in enumerate(user))
    r = post({"uid": "u1", "size": 20, "scene_id": "home_feed", "features": {"user_vector": vec}})
    assert r["code"] == 200 and r["msg"] == "success" and r["size"] == 20 and len(r["items"]) == 20
    assert set(r["items"][0]) == {"item_id", "item_type", "retrieve_id"}
    assert r["items"][0]["retrieve_id"] == "gpu_vector_recall" and r["items"][0]["item_type"] == "video"
    # the same page as the in-process driver
    L = h.L
    direct = json.loads(L.ph_recommend(h.h, b"u1", 20, b"home_feed"))["items"]
    assert [x["item_id"] for x in r["items"]] == [x["item_id"] for x in direct]
    # more than the recall can deliver → 299 "items size not enough"; size <= 0 → 10; unknown user → empty page
    assert post({"uid": "u1", "size": 1000, "scene_id": "home_feed"})["code"] == 299
    assert post({"uid": "u1", "size": 0, "scene_id": "home_feed"})["size"] == 10
    r = post({"uid": "nobody", "size": 5, "scene_id": "home_feed"})
    assert r["code"] == 299 and r["size"] == 0 and r["items"] == []
    assert post({"size": 5})["code"] == 400
    srv.shutdown()
    h.close()


def test_asttype_antlr_outside_the_served_subset_is_refused_at_load(H):
    """RankConf.ASTType = "antlr" selects the valuate evaluator in the reference (GetExpASTWithType / ExprASTResultWithType,
    utils/ast/ast.go:338-389).  The engine serves the subset of that language the reference's tests pin
    (tests/test_expr_antlr.py); a scene whose RankScore or ScoreRewrite uses anything else is an error at load — before any
    GPU work — never a silent evaluation by another grammar."""
    import copy
    cfg = copy.deepcopy(CONFIG)
    cfg["RankConf"]["home_feed"]["ASTType"] = "antlr"
    cfg["RankConf"]["home_feed"]["RankScore"] = "${gpu_dnn} > 0.5 ? 1 : 0"
    assert not H.ph_engine_create(json.dumps(cfg).encode())
    msg = H.ph_last_error()
    assert b'ASTType "antlr"' in msg and b"RankConf[home_feed].RankScore" in msg and b"utils/ast/ast_test.go" in msg
    cfg["RankConf"]["home_feed"]["RankScore"] = "${gpu_dnn}*2"
    cfg["RankConf"]["home_feed"]["ScoreRewrite"] = {"gpu_dnn": "exp(${gpu_dnn})"}
    assert not H.ph_engine_create(json.dumps(cfg).encode())
    assert b"ScoreRewrite[gpu_dnn]" in H.ph_last_error() and b'"exp"' in H.ph_last_error()


@pytest.mark.gpu
def test_multi_output_algorithm_on_one_shared_trunk(H):
    """The same scene served by ONE two-output model (PG_MODEL_DNN3_MULTI, the trunk shared as in the exported PAI-EAS model
    of easyrec_response.go:35-70): one pg_rank_dnn3 per batch writes both "<algo>_<output>" scores; an output list that
    does not match the model is an error of the algorithm (the batch keeps its scores, rank_service.go:274-276)."""
    import copy
    import pairec_amd as pa
    H.ph_engine_load_dnn3_multi.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    cfg = copy.deepcopy(CONFIG)
    cfg["UserDefineConfs"]["pairec_gpu"]["Algorithms"][1] = {"Name": "ppnet", "Kind": "dnn3", "Outputs": ["probs_ctr", "probs_cvr"]}
    expr = "(${ppnet_probs_ctr}+2*${ppnet_probs_cvr})*(1+${current_score})^0.1"
    cfg["RankConf"]["home_feed"] = {"RankAlgoList": ["ppnet"], "RankScore": expr, "BatchCount": 100}
    h, _, user = _engine(H, cfg)
    w = o.Dnn3MultiWeights(2, seed=o.SEED_WEIGHTS ^ 0x33)
    blob = pa.pack_dnn3_multi(w.w1, w.b1, w.w2, w.b2, w.w3m, w.b3m, 128)
    assert H.ph_engine_load_dnn3_multi(h, b"ppnet", pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
    out = json.loads(H.ph_recommend(h, b"u1", 40, b"home_feed"))["items"]
    tab = o.synth_rows(o.SEED_TABLE, 0, 20000, 128)
    rows, scores = o.recall_topk(tab, user[None], 300)
    ref = o.dnn3_multi_forward(w, 0, user, tab[rows[0].astype(np.int64)])
    fused = (o.widen_f32(ref[0]) + 2 * o.widen_f32(ref[1])) * (1 + o.widen_f32(scores[0])) ** 0.1
    assert [x["item_id"] for x in out] == ["item_%d" % rows[0][i] for i in o.sort_scores(fused, True)[:40]]
    pos = {"item_%d" % r: i for i, r in enumerate(rows[0])}
    for x in out:
        i = pos[x["item_id"]]
        a, b = x["algo_scores"]["ppnet_probs_ctr"], x["algo_scores"]["ppnet_probs_cvr"]
        assert abs(a - ref[0][i]) <= 2e-7 and abs(b - ref[1][i]) <= 2e-7 and "ppnet" not in x["algo_scores"]
        assert abs(x["score"] - (a + 2 * b) * (1 + float(scores[0][i])) ** 0.1) <= 1e-12
    # three heads behind a two-output algorithm: the algorithm errs, the items keep no ppnet scores (RankScore then reads 0)
    w3 = o.Dnn3MultiWeights(3)
    blob = pa.pack_dnn3_multi(w3.w1, w3.b1, w3.w2, w3.b2, w3.w3m, w3.b3m, 128)
    assert H.ph_engine_load_dnn3_multi(h, b"ppnet", pa.PREC_F32, blob, len(blob)) == 0
    out = json.loads(H.ph_recommend(h, b"u1", 10, b"home_feed"))["items"]
    assert all("ppnet_probs_ctr" not in x["algo_scores"] for x in out)
    H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_online_hologres_vector_recall(H):
    """OnlineHologresVectorRecall (service/recall/online_hologres_vector_recall.go:105-237): the user's features → the vector
    model's user embedding → the RecallCount rows of the Hologres vector table with the largest inner product, restricted by
    HologresVectorConf.WhereClause; Item.Score = distance, RetrieveId = the recall's name.  Kind "online_hologres": the user
    tower runs on the device, the vector table is the engine's item-embedding table, the clause a feature column."""
    import copy
    import pairec_amd as pa
    H.ph_engine_load_fm2t.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    H.ph_engine_set_feature_column.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    n = 20000
    for coalesce in (False, True):
        cfg = copy.deepcopy(CONFIG)
        g = cfg["UserDefineConfs"]["pairec_gpu"]
        g["OnlineVector"] = {"Rows": n, "Dim": 64, "SyntheticSeed": o.SEED_TABLE ^ 0x77}
        if coalesce:
            g["Coalesce"] = {"MaxWaitUs": 200}
        g["Recalls"] += [{"Name": "oh_all", "Kind": "online_hologres", "RecallCount": 90},
                         {"Name": "oh_cat", "Kind": "online_hologres", "RecallCount": 90,
                          "HologresVectorConf": {"WhereClause": "cat_id = 3"}},
                         {"Name": "oh_nocol", "Kind": "online_hologres", "RecallCount": 90, "WhereClause": "missing > 1"}]
        for name in ("oh_all", "oh_cat", "oh_nocol"):
            cfg["SceneConfs"]["s_" + name] = {"default": {"RecallNames": [name]}}
        h, _, user = _engine(H, cfg)
        fw = o.Fm2tWeights(vocab=500)
        blob = pa.pack_fm2t(fw)
        assert H.ph_engine_load_fm2t(h, pa.PREC_F32, blob, len(blob)) == 0, H.ph_last_error()
        rng = np.random.default_rng(12)
        cat = rng.integers(0, 8, n).astype(np.int32)
        assert H.ph_engine_set_feature_column(h, b"cat_id", cat.ctypes.data, n) == 0, H.ph_last_error()
        emb = o.synth_rows(o.SEED_TABLE ^ 0x77, 0, n, 64)
        ue = o.fm2t_user_embedding(fw, 0, user)
        # unfiltered: the whole vector table
        out = json.loads(H.ph_recommend(h, b"u1", 90, b"s_oh_all"))["items"]
        orow, osc = o.recall_topk(emb, ue[None], 90)
        assert [x["item_id"] for x in out] == ["item_%d" % r for r in orow[0]]
        assert [x["score"] for x in out] == [float(s_) for s_ in osc[0]] and {x["retrieve_id"] for x in out} == {"oh_all"}
        # WHERE cat_id = 3: the oracle on the admitted rows alone; asked twice (the second answer comes from the user-embedding cache)
        keep = np.nonzero(cat == 3)[0]
        frow, fsc = o.recall_topk(emb[keep], ue[None], 90)
        for _ in range(2):
            out = json.loads(H.ph_recommend(h, b"u1", 90, b"s_oh_cat"))["items"]
            assert [x["item_id"] for x in out] == ["item_%d" % keep[r] for r in frow[0]]
            assert [x["score"] for x in out] == [float(s_) for s_ in fsc[0]] and {x["retrieve_id"] for x in out} == {"oh_cat"}
        # an unknown column: the reference's SQL fails — logged, empty list
        assert json.loads(H.ph_recommend(h, b"u1", 90, b"s_oh_nocol"))["items"] == []
        H.ph_engine_destroy(h)


@pytest.mark.gpu
def test_feature_columns_change_over_with_the_table_generation(H):
    """A new generation of the table brings its own row-keyed feature columns (ph_engine_ingest_feature_column between begin
    and commit): rows, ids and columns change over in ONE exclusive section, so a `WhereClause` never reads the previous
    generation's column against the new rows; a generation that brings none while the engine serves columns is refused."""
    import copy
    H.ph_engine_set_feature_column.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    H.ph_engine_ingest_begin.argtypes = [C.c_void_p]
    H.ph_engine_ingest_chunk.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint64]
    H.ph_engine_ingest_feature_column.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    H.ph_engine_ingest_commit.argtypes = [C.c_void_p]
    H.ph_ingest_last_error.restype = C.c_char_p
    n = 20000
    cfg = copy.deepcopy(CONFIG)
    g = cfg["UserDefineConfs"]["pairec_gpu"]
    g["Coalesce"] = {"MaxWaitUs": 200}
    g["Recalls"].append({"Name": "holo_cat", "Kind": "hologres", "RecallCount": 60, "WhereClause": "cat_id = 2"})
    cfg["SceneConfs"]["s_cat"] = {"default": {"RecallNames": ["holo_cat"]}}
    h, _, user = _engine(H, cfg)
    rng = np.random.default_rng(21)
    tab_a = o.synth_rows(o.SEED_TABLE, 0, n, 128)
    cat_a = rng.integers(0, 5, n).astype(np.int32)
    assert H.ph_engine_set_feature_column(h, b"cat_id", cat_a.ctypes.data, n) == 0

    def page():
        return json.loads(H.ph_recommend(h, b"u1", 60, b"s_cat"))["items"]

    def want(tab, cat, prefix):
        keep = np.nonzero(cat == 2)[0]
        r, s_ = o.recall_topk(tab[keep], user[None], 60)
        return sorted("%s%d" % (prefix, keep[i]) for i in r[0])
    assert sorted(x["item_id"] for x in page()) == want(tab_a, cat_a, "item_")
    # generation B: other rows, other ids, another column
    tab_b = o.synth_rows(o.SEED_TABLE ^ 0x5, 0, n, 128)
    cat_b = rng.integers(0, 5, n).astype(np.int32)
    ids_b = b"".join(b"b%d\0" % i for i in range(n))
    assert H.ph_engine_ingest_begin(h) == 0
    assert H.ph_engine_ingest_chunk(h, ids_b, len(ids_b), tab_b.ctypes.data, n) == 0
    assert H.ph_engine_ingest_commit(h) != 0 and b"must bring its own" in H.ph_ingest_last_error()      # no column staged
    assert sorted(x["item_id"] for x in page()) == want(tab_a, cat_a, "item_")                           # still generation A, whole
    assert H.ph_engine_ingest_feature_column(h, b"cat_id", cat_b.ctypes.data, n) == 0, H.ph_ingest_last_error()
    assert H.ph_engine_ingest_commit(h) == 0, H.ph_ingest_last_error()
    assert sorted(x["item_id"] for x in page()) == want(tab_b, cat_b, "b")
    # an id buffer with a NUL inside an id is refused (it would shift every later id by a row)
    assert H.ph_engine_ingest_begin(h) == 0
    bad = b"x\0y\0" + b"".join(b"c%d\0" % i for i in range(n - 1))
    assert H.ph_engine_ingest_chunk(h, bad, len(bad), tab_b.ctypes.data, n) != 0 and b"more than one id per row" in H.ph_ingest_last_error()
    H.ph_engine_destroy(h)
