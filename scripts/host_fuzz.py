"""Mutation fuzz of the host mirror's parse / format entry points — recconf, recall-conf rules, response decoders, UniqueFilter
input, cache lines, EasyRec generator spec, vector strings, Go float formatting, and (round 5's additions) the normalizer /
feature-operator specs with their govaluate / expr-lang subsets — for a given number of seconds.  CPU only; run by
scripts/host_asan.sh against the AddressSanitizer + UBSan build.  Usage: host_fuzz.py [seconds] [seed]"""
import ctypes as C, json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.environ.get("PH_HOST_LIB") or os.path.join(ROOT, "pairec_amd", "libpairec_host.so"))
one = ("ph_parse_recconf", "ph_check_recall_conf", "ph_decode_response", "ph_unique_filter", "ph_easyrec_generator",
       "ph_normalizer_apply", "ph_feature_load")
for f in one:
    getattr(L, f).restype = C.c_char_p
    getattr(L, f).argtypes = [C.c_char_p]
L.ph_format_recall_cache.restype = C.c_char_p
L.ph_format_recall_cache.argtypes = [C.c_char_p, C.c_char_p]
L.ph_parse_recall_cache.restype = C.c_char_p
L.ph_parse_recall_cache.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
L.ph_parse_vector_string.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int]
L.ph_go_fmt_float.restype = C.c_char_p
L.ph_go_fmt_float.argtypes = [C.c_double]
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sys.path.insert(0, ROOT)
from tests.test_host_mirror import CONFIG
seeds = [json.dumps(CONFIG),
         json.dumps({"Name": "r", "RecallType": "VectorRecall", "DaoConf": {"AdapterType": "redis"}}),
         json.dumps([{"id": "1", "score": 0.5, "retrieve_id": "r1", "algo_scores": {"m": 0.7}}, {"id": "2", "score": 0.4, "retrieve_id": "r1", "algo_scores": {}}]),
         json.dumps({"func": "easyrecMutValResponseFunc", "item_ids": ["a", "zz"], "outputs": ["probs_ctr", "probs_cvr"], "results": {"a": [0.11, 0.0069]}}),
         json.dumps({"context_features": ["f"], "item_features": None, "user": {"u": 1}, "items": [{"id": "1", "features": {"f": 2}}], "batches": [1]}),
         json.dumps({"normalizer": "expression", "expression": "log(x + 1) * 2 > 1 ? hash32(y) : geoHash(1.5, 2.5, 6)", "values": [1, 2.5, "a", None]}),
         json.dumps({"normalizer": "expr", "expression": "s2CellID(lat, lng, 12) % 7 + len(name)", "values": [{"lat": 30.1, "lng": 120.2, "name": "x"}]}),
         json.dumps({"features": [{"FeatureName": "f", "FeatureType": "new_feature", "FeatureSource": "item:price", "Normalizer": "hour_in_day"},
                                  {"FeatureName": "g", "FeatureType": "raw_feature", "FeatureSource": "user:age", "Normalizer": "random", "FeatureValue": "3"}],
                     "user": {"age": 31}, "items": [{"id": "1", "price": 1700000000}]}),
         "1:0.12 2:-0.3 junk 3:1e-2 4:x 5:1:2", "item_1:recall:0.5,item_2:recall:0.25,item3", "create_time > ${time}"]
alphabet = list('{}[]":,.-+eE0123456789 \\\t\n/ntfalsrue$()?<>=!&|%*') + ['\xff', 'é', '"', '\\u12', '\\"', 'log', 'hash', '1e999']


def mutate(s):
    s = list(s)
    for _ in range(rnd.randint(1, 6)):
        k = rnd.randint(0, 4)
        i = rnd.randrange(len(s) + 1)
        if k == 0 and s:
            del s[min(i, len(s) - 1)]
        elif k == 1:
            s.insert(i, rnd.choice(alphabet))
        elif k == 2 and s:
            s[min(i, len(s) - 1)] = rnd.choice(alphabet)
        elif k == 3:
            s = s[:i]
        else:
            s[i:i] = s[max(0, i - rnd.randint(1, 30)):i]
    return "".join(s)


buf = (C.c_float * 8)()
n, t_end = 0, time.time() + seconds
while time.time() < t_end:
    for _ in range(200):
        b = mutate(rnd.choice(seeds)).encode("utf-8", "ignore").replace(b"\x00", b"")
        for f in one:
            getattr(L, f)(b)
        L.ph_format_recall_cache(b, b"r")
        L.ph_parse_recall_cache(b, b"r", b"t")
        L.ph_parse_vector_string(b, buf, rnd.randint(0, 8))
        L.ph_go_fmt_float(rnd.choice([0.0, -0.0, 1e21, 1e-7, 123456789.125, float("inf"), float("nan"), rnd.uniform(-1e9, 1e9),
                                      rnd.random() * 10 ** rnd.randint(-30, 30)]))
        n += 1
print("host_fuzz: %d mutated inputs through %d host parsers / formatters in %.0f s: no sanitizer report" % (n, len(one) + 4, seconds))
