"""Re-encodes the known-answer tests the reference holds for the hot path as data.

Nothing is imported from /root/reference (it is Go; there is no Go toolchain here): the constants
below were transcribed by hand from the cited test functions, which are the only reference-side
golden data that exists for this path.  Run:  python tests/golden/make_reference_known_answers.py
"""
import json, os

cases = {
    "_source": "alibaba/pairec known-answer tests, transcribed as data (inputs + expected outputs only)",
    "expr": [
        {"ref": "utils/ast/ast_test.go:13-29 TestAST", "expr": "${ctr} + ${click} + ${price}",
         "algo_scores": {"ctr": 0.1, "click": 0.3}, "properties": {"price": 0.1}, "expect": 0.5},
        {"ref": "utils/ast/ast_test.go:90-129 TestGoAntlrEvaluate3",
         "expr": "(${ppnet_probs_ctr}+2*${ppnet_probs_cvr})*${log_price}^0.1",
         "algo_scores": {"ppnet_probs_ctr": 0.11173942685127258, "ppnet_probs_cvr": 0.006906657014042139},
         "properties": {"log_price": 1.0986122886681098},
         "expect_formula": "(a+2*b)*pow(c,0.1)", "formula_args": ["ppnet_probs_ctr", "ppnet_probs_cvr", "log_price"]},
        {"ref": "utils/ast/ast_test.go:131-167 TestGoAntlrEvaluate4",
         "expr": "(${cdn_probs_ctr}+2*${cdn_probs_cvr})*${log_price}^0.1",
         "algo_scores": {"cdn_probs_ctr": 0.004212516359984875, "cdn_probs_cvr": 0.0014530600747093558},
         "properties": {"log_price": 0.6931471805599453},
         "expect_formula": "(a+2*b)*pow(c,0.1)", "formula_args": ["cdn_probs_ctr", "cdn_probs_cvr", "log_price"]},
        {"ref": "utils/ast/ast_test.go:213-237 TestASTWithType (ASTType \"\")", "expr": "${ctr} + ${click} + ${price}",
         "algo_scores": {"ctr": 0.1, "click": 0.3}, "properties": {"price": 0.1}, "expect": 0.5},
        {"ref": "sort/boost_score_sort_test.go:13-70 (score*100 on item 1)", "expr": "${score} * 100",
         "algo_scores": {}, "properties": {"score": 1.0}, "expect": 100.0},
        {"ref": "sort/boost_score_sort_test.go:13-70 (score*(-10) on item 10)", "expr": "${score} * (-10)",
         "algo_scores": {}, "properties": {"score": 10.0}, "expect": -100.0},
        {"ref": "sort/boost_score_sort_test.go:13-70 (score*100 on item 0)", "expr": "${score} * 100",
         "algo_scores": {}, "properties": {"score": 0.0}, "expect": 0.0},
    ],
    "sort": [
        {"ref": "sort/multi_recall_mix_sort_test.go:27-50 (20 items, Score=i, ItemRankScoreSort)",
         "scores": [float(i) for i in range(20)], "descending": True,
         "expect_order": list(range(19, -1, -1))},
        {"ref": "sort/item_score.go:36-41 ItemScoreSort ascending on the same items",
         "scores": [float(i) for i in range(20)], "descending": False,
         "expect_order": list(range(20))},
    ],
    # request-side feature typing (the `features` object of POST /api/recommend)
    "features_map": [
        {"ref": "web/features_map_test.go:16-20 int64 array to string array",
         "json": '{"cellIDNeighbors":[384307167844368384,1921535839221841920,1921535841369325568]}',
         "key": "cellIDNeighbors", "type": "[]string"},
        {"ref": "web/features_map_test.go:22-26 float64 array to string array", "json": '{"scores":[1.5,2.5,3.5]}',
         "key": "scores", "type": "[]string", "value": ["1.5", "2.5", "3.5"]},
        {"ref": "web/features_map_test.go:28-32", "json": '{"name":"test","count":10,"tags":["a","b"]}',
         "key": "tags", "type": "[]string", "value": ["a", "b"]},
        {"ref": "web/features_map_test.go:34-38", "json": '{"name":"test","count":10,"tags":[["a","b"],["c","d"]]}',
         "key": "tags", "type": "[][]string", "value": [["a", "b"], ["c", "d"]]},
        {"ref": "web/features_map_test.go:40-44", "json": '{"name":"test","count":10,"tags":[[1,2],[3,4]]}',
         "key": "tags", "type": "[][]string", "value": [["1", "2"], ["3", "4"]]},
        {"ref": "web/features_map_test.go:46-50 int_field", "json": '{"name":"test","count":10,"tags":[[1,2],[3,4]]}',
         "key": "count", "type": "int", "value": 10},
        {"ref": "web/features_map_test.go:52-56 float_field", "json": '{"name":"test","count":10.2,"tags":[[1,2],[3,4]]}',
         "key": "count", "type": "float64", "value": 10.2},
        {"ref": "web/features_map_test.go:95-127 TestFeaturesMap_PrecisionPreservation",
         "json": '{"cellIDNeighbors":[384307167844368384,1921535839221841920,1921535841369325568,1152921512123039744]}',
         "key": "cellIDNeighbors", "type": "[]string",
         "value": ["384307167844368384", "1921535839221841920", "1921535841369325568", "1152921512123039744"]},
        {"ref": "web/features_map_test.go:129-163 TestFeaturesMap_StringArrayConversion",
         "json": '{"geoHashWithNeighbors":["s00000000001","s00000000003","s00000000002","kpbpbpbpbpbr","kpbpbpbpbpbp","7zzzzzzzzzzz","ebpbpbpbpbpb","ebpbpbpbpbpc","s00000000000"]}',
         "key": "geoHashWithNeighbors", "type": "[]string",
         "value": ["s00000000001", "s00000000003", "s00000000002", "kpbpbpbpbpbr", "kpbpbpbpbpbp", "7zzzzzzzzzzz",
                   "ebpbpbpbpbpb", "ebpbpbpbpbpc", "s00000000000"]},
        {"ref": "web/features_map_test.go:173-177", "json": '{"metadata":{"key1":"value1","key2":"value2"}}',
         "key": "metadata", "type": "map[string]string"},
        {"ref": "web/features_map_test.go:179-183", "json": '{"counts":{"a":1,"b":2,"c":3}}',
         "key": "counts", "type": "map[string]string"},
        {"ref": "web/features_map_test.go:185-189", "json": '{"scores":{"x":1.5,"y":2.5,"z":3.5}}',
         "key": "scores", "type": "map[string]string"},
        {"ref": "web/features_map_test.go:191-195", "json": '{"config":{"name":"test","count":10,"rate":0.5}}',
         "key": "config", "type": "map[string]string"},
        {"ref": "web/features_map_test.go:197-201", "json": '{"tags":{"ids":[1,2,3],"nums":[10,20,30]}}',
         "key": "tags", "type": "map[string][]string"},
        {"ref": "web/features_map_test.go:203-207", "json": '{"names":{"first":["a","b"],"last":["c","d"]}}',
         "key": "names", "type": "map[string][]string"},
        {"ref": "web/features_map_test.go:209-213", "json": '{"nested":{"inner":{"key":"value"}}}',
         "key": "nested", "type": "map[string]interface {}"},
    ],
    "decode": [
        {"ref": "algorithm/eas/easyrec_response_test.go:44-72 (FloatVal [1,6] per item, value [1][5])",
         "float_val": [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.11, 0.22, 0.33, 0.44, 0.55, 0.66], "dim1": 6,
         "item": 1, "index": 5, "expect_f32": 0.66},
        {"ref": "algorithm/eas/fm_response.go:28-34 alinkFMResponse.GetScore", "label": 0.0, "score": 0.8,
         "expect_1_minus": True},
    ],
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.json")
with open(out, "w") as f:
    json.dump(cases, f, indent=1)
print("wrote", out)
