"""Several threads at once through the host mirror's entry points that touch shared state — the registries
(ph_registry_semantics), the normalizers incl. RandomNormalizer's generator and the test clock (ph_normalizer_apply,
ph_feature_load), the recconf parser, UniqueFilter and the float formatter with their thread-local result buffers — for
scripts/host_tsan.sh (ThreadSanitizer build; ctypes calls release the GIL, so the calls really overlap).
Usage: host_threads.py [threads] [seconds]"""
import ctypes as C, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.environ.get("PH_HOST_LIB") or os.path.join(ROOT, "pairec_amd", "libpairec_host.so"))
for f in ("ph_parse_recconf", "ph_unique_filter", "ph_normalizer_apply", "ph_feature_load", "ph_decode_response"):
    getattr(L, f).restype = C.c_char_p
    getattr(L, f).argtypes = [C.c_char_p]
L.ph_go_fmt_float.restype = C.c_char_p
L.ph_go_fmt_float.argtypes = [C.c_double]
sys.path.insert(0, ROOT)
from tests.test_host_mirror import CONFIG
n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
conf = json.dumps(CONFIG).encode()
items = json.dumps([{"id": str(i % 7), "score": 0.5, "retrieve_id": "r%d" % (i % 3), "algo_scores": {"m": 0.1 * i}} for i in range(40)]).encode()
norms = [json.dumps({"name": n, "expression": e, "value": v, "clock_ms": 1700000000000}).encode() for n, e, v in (
    ("random", "", 5), ("hour_in_day", "", 1700000000), ("weekday", "", 1700000000), ("const_value", "", 3),
    ("expression", "x * 2 + 1", 4.5), ("month", "", 1700000000))]
feat = json.dumps({"features": [{"FeatureName": "f", "FeatureType": "new_feature", "FeatureSource": "item:price", "Normalizer": "hour_in_day"},
                                {"FeatureName": "g", "FeatureType": "new_feature", "FeatureSource": "user:age", "Normalizer": "random"}],
                   "user": {"id": "u", "properties": {"age": 31}}, "items": [{"id": "1", "properties": {"price": 1700000000}}],
                   "clock_ms": 1700000000000}).encode()
resp = json.dumps({"func": "easyrecResponseFunc", "item_ids": ["a", "missing", "b"], "results": {"a": [0.25, 9], "b": [0.75]}}).encode()
stop = time.time() + seconds
counts = [0] * n_threads
bad = []


def work(t):
    i = 0
    while time.time() < stop:
        if L.ph_registry_semantics() != 31:
            bad.append("registry semantics")
        a = L.ph_parse_recconf(conf)
        b = L.ph_unique_filter(items)
        c = L.ph_normalizer_apply(norms[(i + t) % len(norms)])
        d = L.ph_feature_load(feat)
        e = L.ph_decode_response(resp)
        g = L.ph_go_fmt_float(1.0 + t + i * 1e-3)
        if not (a and b and c and d and e and g):
            bad.append("a call returned NULL")
        if float(g) != 1.0 + t + i * 1e-3:
            bad.append("float formatter: another thread's buffer")
        i += 1
    counts[t] = i


ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
for th in ths:
    th.start()
for th in ths:
    th.join()
if bad:
    print("host_threads: FAILED:", sorted(set(bad)))
    sys.exit(1)
print("host_threads: %d threads x %.0f s, %d rounds of 7 entry points: consistent" % (n_threads, seconds, sum(counts)))
