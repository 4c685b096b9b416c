"""Mutation fuzz of the host mirror's parse / format entry points (recconf, recall-conf rules, response decoders, UniqueFilter
input, cache lines, EasyRec generator spec, vector strings, Go float formatting) under AddressSanitizer + UBSan — CPU only.
Built and run by scripts/fuzz/run_host_asan.sh.  Usage: fuzz_host.py [seed] [count]"""
import ctypes as C, json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.environ.get('PH_ASAN_LIB', '/tmp/pairec_asan/libpairec_host_asan.so'))
L.ph_last_error.restype = C.c_char_p
for f in ("ph_parse_recconf", "ph_check_recall_conf", "ph_decode_response", "ph_unique_filter", "ph_format_recall_cache", "ph_parse_recall_cache", "ph_easyrec_generator"):
    getattr(L, f).restype = C.c_char_p
L.ph_parse_recconf.argtypes = [C.c_char_p]; L.ph_check_recall_conf.argtypes = [C.c_char_p]; L.ph_decode_response.argtypes = [C.c_char_p]
L.ph_unique_filter.argtypes = [C.c_char_p]; L.ph_format_recall_cache.argtypes = [C.c_char_p, C.c_char_p]
L.ph_parse_recall_cache.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]; L.ph_easyrec_generator.argtypes = [C.c_char_p]
L.ph_parse_vector_string.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int]
L.ph_go_fmt_float.restype = C.c_char_p; L.ph_go_fmt_float.argtypes = [C.c_double]
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
sys.path.insert(0, ROOT)
from tests.test_host_mirror import CONFIG
seeds = [json.dumps(CONFIG),
         json.dumps({"Name": "r", "RecallType": "VectorRecall", "DaoConf": {"AdapterType": "redis"}}),
         json.dumps([{"id": "1", "score": 0.5, "retrieve_id": "r1", "algo_scores": {"m": 0.7}}, {"id": "2", "score": 0.4, "retrieve_id": "r1", "algo_scores": {}}]),
         json.dumps({"kind": "easyrec", "item_ids": ["a", "b"], "results": {"a": [0.5], "b": [0.25, 0.1]}}),
         json.dumps({"context_features": ["f"], "item_features": None, "user": {"u": 1}, "items": [{"id": "1", "features": {"f": 2}}], "batches": [1]}),
         "1:0.12 2:-0.3 junk 3:1e-2 4:x 5:1:2", "item_1:recall:0.5,item_2:recall:0.25,item3", "create_time > ${time}"]
alphabet = list('{}[]":,.-+eE0123456789 \\\t\n/ntfalsrue$') + ['\x00', '\xff', 'é', '"', '\\u12', '\\"']
def mutate(s):
    s = list(s)
    for _ in range(rnd.randint(1, 6)):
        k = rnd.randint(0, 4)
        i = rnd.randrange(len(s) + 1)
        if k == 0 and s: del s[min(i, len(s) - 1)]
        elif k == 1: s.insert(i, rnd.choice(alphabet))
        elif k == 2 and s: s[min(i, len(s) - 1)] = rnd.choice(alphabet)
        elif k == 3: s = s[:i]
        else: s[i:i] = s[max(0, i - rnd.randint(1, 30)):i]
    return "".join(s)
buf = (C.c_float * 8)()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
for it in range(n):
    src = mutate(rnd.choice(seeds))
    b = src.encode('utf-8', 'ignore').replace(b'\x00', b'')
    L.ph_parse_recconf(b); L.ph_check_recall_conf(b); L.ph_decode_response(b); L.ph_unique_filter(b); L.ph_easyrec_generator(b)
    L.ph_format_recall_cache(b, b"r"); L.ph_parse_recall_cache(b, b"r", b"t"); L.ph_parse_vector_string(b, buf, rnd.randint(0, 8))
    L.ph_go_fmt_float(rnd.choice([0.0, -0.0, 1e21, 1e-7, 123456789.125, float('inf'), float('nan'), rnd.uniform(-1e9, 1e9), rnd.random() * 10 ** rnd.randint(-30, 30)]))
print("fuzz_host: %d mutated inputs through 9 host parsers / formatters: no sanitizer report" % n)
