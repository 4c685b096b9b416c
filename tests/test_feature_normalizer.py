"""service/feature in the host mirror (pairec_amd/host/feature.cpp): NewNormalizer's named normalizers, the
subsets of the two expression languages, utils.GovaluateFunctions and the FeatureOp family behind Feature.LoadFeatures —
pinned by the reference's own tests (tests/golden/reference_known_answers.json "feature_normalizer" / "feature_load", transcribed
from service/feature/{normalizer,feature}_test.go) and, for the hashes, by independent implementations (python-xxhash; the
published MurmurHash3 test vectors).  CPU only: nothing here touches the device."""
import ctypes as C
import datetime
import json
import math
import os
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))


@pytest.fixture(scope="module")
def H():
    # (PH_HOST_LIB: scripts/host_asan.sh / host_tsan.sh point the same tests at the sanitizer builds of the library)
    L = C.CDLL(os.environ.get("PH_HOST_LIB") or os.path.join(ROOT, "pairec_amd", "libpairec_host.so"))
    L.ph_last_error.restype = C.c_char_p
    L.ph_normalizer_apply.restype = C.c_char_p
    L.ph_normalizer_apply.argtypes = [C.c_char_p]
    L.ph_feature_load.restype = C.c_char_p
    L.ph_feature_load.argtypes = [C.c_char_p]
    return L


def apply(H, name, expression="", value=None, **kw):
    r = H.ph_normalizer_apply(json.dumps(dict(name=name, expression=expression, value=value, **kw)).encode())
    if r is None:
        raise ValueError(H.ph_last_error().decode())
    return json.loads(r)


def load(H, spec):
    r = H.ph_feature_load(json.dumps(spec).encode())
    if r is None:
        raise ValueError(H.ph_last_error().decode())
    return json.loads(r)


def murmur3_32(data: bytes, seed: int = 0) -> int:
    """MurmurHash3_x86_32 (Appleby's reference algorithm), an independent restatement for the test."""
    c1, c2, h, n = 0xcc9e2d51, 0x1b873593, seed, len(data)
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & 0xffffffff
    for i in range(0, n - n % 4, 4):
        k = int.from_bytes(data[i:i + 4], "little")
        k = (k * c1) & 0xffffffff; k = rotl(k, 15); k = (k * c2) & 0xffffffff
        h ^= k; h = rotl(h, 13); h = (h * 5 + 0xe6546b64) & 0xffffffff
    k = 0
    tail = data[n - n % 4:]
    for j in range(len(tail) - 1, -1, -1):
        k ^= tail[j] << (8 * j)
    if tail:
        k = (k * c1) & 0xffffffff; k = rotl(k, 15); k = (k * c2) & 0xffffffff
        h ^= k
    h ^= n
    h ^= h >> 16; h = (h * 0x85ebca6b) & 0xffffffff; h ^= h >> 13; h = (h * 0xc2b2ae35) & 0xffffffff; h ^= h >> 16
    return h


def test_murmur3_restatement_against_published_vectors():
    # the vectors every MurmurHash3 port is checked with (seed 0)
    assert murmur3_32(b"") == 0
    assert murmur3_32(b"hello") == 0x248bfa47
    assert murmur3_32(b"hello, world") == 0x149bbb7f
    assert murmur3_32(b"The quick brown fox jumps over the lazy dog") == 0x2e4ff723


def test_reference_normalizer_known_answers(H):
    fns = {"log": lambda v: math.log(v["a"]), "log10": lambda v: math.log10(v["a"]), "log2": lambda v: math.log2(v["a"]),
           "murmur3_32_mod_100": lambda v: murmur3_32(v["key"].encode()) % 100}
    n = 0
    for g in GOLD["feature_normalizer"]:
        for c in g["cases"]:
            r = apply(H, g["name"], g["expression"], c["value"])
            assert r["type"] == c["expect_type"], (g["ref"], r)
            if "expect_fn" in c:
                assert r["result"] == fns[c["expect_fn"]](c["value"]), (g["ref"], r)
            elif "expect_trunc" in c:
                assert int(r["result"]) == c["expect_trunc"], (g["ref"], r)
            else:
                assert r["result"] == c["expect"] and type(r["result"]) is type(c["expect"]), (g["ref"], r)
            n += 1
    assert n >= 24


def test_reference_feature_load_known_answers(H):
    for g in GOLD["feature_load"]:
        out = load(H, {k: g[k] for k in ("features", "user", "items")})
        for got, want in zip(out["items"], g.get("expect_items", [])):
            for k, v in want.items():
                assert got[k] == v and type(got[k]) is type(v), (g["ref"], got)
        for k, v in g.get("expect_user", {}).items():
            assert out["user"][k] == v and type(out["user"][k]) is type(v), (g["ref"], out["user"])


def test_time_normalizers_follow_the_local_clock(H):
    """hour_in_day / weekday (Monday 0) / month / week (ISO 8601) over pinned instants: every day of seven years' turns, where
    the ISO week differs from the naive one, and a spread of hours."""
    instants = []
    for year in range(2019, 2033):
        for day in list(range(-6, 8)) + [59, 60, 180]:
            instants.append(datetime.datetime(year, 1, 1, 12, 30) + datetime.timedelta(days=day))
    instants += [datetime.datetime(2026, 10, 2, h, 59, 59) for h in range(24)]
    for dt in instants:
        ms = int(time.mktime(dt.timetuple())) * 1000
        lt = time.localtime(ms // 1000)
        d = datetime.date(lt.tm_year, lt.tm_mon, lt.tm_mday)
        assert apply(H, "hour_in_day", clock_ms=ms) == {"kind": "CreateHourNormalizer", "type": "int", "result": lt.tm_hour}
        assert apply(H, "weekday", clock_ms=ms)["result"] == lt.tm_wday              # normalizer.go:51-68: Monday 0 … Sunday 6
        assert apply(H, "month", clock_ms=ms)["result"] == lt.tm_mon
        assert apply(H, "week", clock_ms=ms)["result"] == d.isocalendar()[1], dt
    # the system clock when none is pinned
    before = time.localtime()
    got = apply(H, "hour_in_day")["result"]
    assert got in (before.tm_hour, time.localtime().tm_hour)
    # timestamp() / timestamp('ms') (govaluate_functions.go:258-267)
    t0 = time.time()
    s = apply(H, "expression", "timestamp()", {})["result"]
    ms = apply(H, "expression", "timestamp('MS')", {})["result"]
    assert math.floor(t0) <= s <= math.floor(time.time()) + 1 and math.floor(t0 * 1000) <= ms <= time.time() * 1000 + 1
    assert apply(H, "expression", "timestamp('ms')", {}, clock_ms=1700000000123)["result"] == 1700000000123.0


def test_random_const_and_unknown_normalizers(H):
    seen = {apply(H, "random")["result"] for _ in range(400)}
    assert seen <= set(range(100)) and len(seen) > 50                               # rand.Intn(100)
    assert apply(H, "const_value") == {"kind": "CreateConstValueNormalizer", "type": "nil", "result": None}
    assert apply(H, "no_such_normalizer") == {"kind": None}                        # normalizer.go:21-40: the interface stays nil
    assert apply(H, "") == {"kind": None}


def test_hash_functions_against_independent_implementations(H):
    xxhash = pytest.importorskip("xxhash")
    keys = ["", "a", "hello world", "retarget_u2i", "x" * 31, "y" * 32, "z" * 33, "w" * 100, "中文特征", "0123456789abcdef" * 5]
    for k in keys:
        r = apply(H, "expression", "hash(k)", {"k": k})
        assert r["type"] == "uint64" and r["result"] == xxhash.xxh64(k.encode()).intdigest(), k
        r = apply(H, "expression", "hash32(k)", {"k": k})
        assert r["type"] == "float64" and r["result"] == float(murmur3_32(k.encode())), k


def test_geo_functions(H):
    p = {"lat": 39.9042, "lng": 116.4074}
    # prefixes of one cell: every precision is the 12-character hash cut short
    full = apply(H, "expression", "geoHash(lat, lng, 12)", p)["result"]
    assert len(full) == 12 and full.startswith("wx4g0b")
    for n in range(1, 13):
        assert apply(H, "expression", "geoHash(lat, lng, %d)" % n, p)["result"] == full[:n]
    # an independent decode of the hash brackets the point
    alphabet = "0123456789bcdefghjkmnpqrstuvwxyz"
    lat, lng, even = [-90.0, 90.0], [-180.0, 180.0], True
    for ch in full:
        v = alphabet.index(ch)
        for b in (16, 8, 4, 2, 1):
            rng = lng if even else lat
            mid = (rng[0] + rng[1]) / 2
            if v & b:
                rng[0] = mid
            else:
                rng[1] = mid
            even = not even
    assert lat[0] <= 39.9042 <= lat[1] and lng[0] <= 116.4074 <= lng[1]
    # well-known cells
    assert apply(H, "expression", "geoHash(a, b, 11)", {"a": 57.64911, "b": 10.40744})["result"] == "u4pruydqqvj"
    assert apply(H, "expression", "geoHash(a, b, 5)", {"a": 0, "b": 0})["result"] == "s0000"
    # s2: a parent's id is its child's with the trailing bits folded (level 20 vs 15), the face in the top three bits
    c15 = apply(H, "expression", "s2CellID(lat, lng)", p)["result"]
    c20 = apply(H, "expression", "s2CellID(lat, lng, 20)", p)["result"]
    lsb15 = 1 << (2 * (30 - 15))
    assert c15 == 3886697436164390912 and (c20 & -lsb15) | lsb15 == c15 and c20 & -c20 == 1 << (2 * (30 - 20))
    assert apply(H, "expression", "s2CellID(a, b, 0)", {"a": 0, "b": 0})["result"] == 0x1000000000000000        # face 0
    assert apply(H, "expression", "s2CellID(a, b, 0)", {"a": 90, "b": 0})["result"] == 0x5000000000000000       # face 2: the north pole
    assert apply(H, "expression", "s2CellID(a, b, 0)", {"a": 0, "b": 90})["result"] == 0x3000000000000000       # face 1
    # out-of-range coordinates are an evaluation error, i.e. "" (normalizer.go:126-138)
    assert apply(H, "expression", "geoHash(a, b)", {"a": 90, "b": 0})["result"] == ""
    d = apply(H, "expression", "haversine(lng1, lat1, lng2, lat2)", {"lat1": 10, "lng1": 20, "lat2": 10, "lng2": 20})["result"]
    assert d == 0.0


def test_string_functions_and_to_string(H):
    e = lambda x, v: apply(H, "expression", x, v)["result"]
    assert e("getString(a, b)", {"a": "", "b": "fallback"}) == "fallback"
    assert e("getString(a)", {"a": ""}) == ""
    assert e("getString(a, b)", {"a": 5, "b": "x"}) == 5.0                          # a number is not "": it is returned as it is
    assert e("trim(a, ' _')", {"a": "__ ab _c _ "}) == "ab _c"
    assert e("trim(a, '中')", {"a": "中中文中"}) == "文"
    assert e("trimPrefix(a, 'pre_')", {"a": "pre_pre_x"}) == "pre_x"
    assert e("trimPrefix(a, 'zz')", {"a": "pre_x"}) == "pre_x"
    assert e("replace(a, '_', '-')", {"a": "a_b__c"}) == "a-b--c"
    assert e("replace(a, '', '.')", {"a": "ab"}) == ".a.b."                         # strings.ReplaceAll with an empty old
    assert e("replace(a, 1.5, 2)", {"a": "x1.5y"}) == "x2y"                         # utils.ToString: FormatFloat 'f', -1
    assert e("round(a)", {"a": 2.5}) == 3.0 and e("round(a)", {"a": -2.5}) == -3.0  # math.Round: half away from zero
    assert e("round(a, 2)", {"a": 3.14159}) == 3.14 and e("round(a, 2)", {"a": 2.999}) == 2.99   # two arguments TRUNCATE
    assert e("toFloat64(a) + 1", {"a": "41.5"}) == 42.5
    assert e("toFloat64(a)", {"a": "oops"}) == 0.0
    assert e("'n=' + a", {"a": 1000000}) == "n=1e+06"                               # fmt %v of a float64
    assert e("'n=' + a", {"a": 0.5}) == "n=0.5"
    assert e("a + b", {"a": "x", "b": True}) == "xtrue"
    # arity errors are evaluation errors
    assert e("max(a)", {"a": 1}) == "" and e("haversine(a, a)", {"a": 1}) == "" and e("maxIndex(a)", {"a": []}) == ""


def test_govaluate_subset_semantics(H):
    e = lambda x, v=None: apply(H, "expression", x, v if v is not None else {})
    assert e("1 + 2 * 3 - 4 / 2")["result"] == 5.0
    assert e("(1 + 2) * 3")["result"] == 9.0
    assert e("7 % 4")["result"] == 3.0 and e("-7 % 4")["result"] == -3.0 and e("7.5 % 2")["result"] == 1.5    # math.Mod
    assert e("2 ** 10")["result"] == 1024.0
    assert e("-2 ** 2")["result"] == 4.0                                             # the prefix binds first: (-2) ** 2
    assert e("0x10 + 1")["result"] == 17.0
    assert e("a", {"a": 3})["type"] == "float64"                                     # every integer parameter becomes a float64
    assert e("a == 3", {"a": 3})["result"] is True
    assert e("a == '3'", {"a": 3})["result"] is False                                # DeepEqual across kinds
    assert e("a != b", {"a": "x", "b": "y"})["result"] is True
    assert e("a < b", {"a": "abc", "b": "abd"})["result"] is True                    # strings compare with strings
    assert e("a < b", {"a": "abc", "b": 1})["result"] == ""                          # … not with numbers: an error, i.e. ""
    assert e("a in (1, 2, 3)", {"a": 2})["result"] is True
    assert e("a in (1, 2, 3)", {"a": 5})["result"] is False
    assert e("a in (1)", {"a": 1})["result"] == ""                                   # (1) is not an array
    assert e("a in b", {"a": 2, "b": [1, 2]})["result"] is True                      # an array parameter
    assert e("a > 1 && b < 1", {"a": 2, "b": 0})["result"] is True
    assert e("a > 1 || nosuch > 1", {"a": 2})["result"] is True                      # short circuit: the right side is not evaluated
    assert e("a > 5 || nosuch > 1", {"a": 2})["result"] == ""                        # … here it is, and the parameter is missing
    assert e("!(a > 1)", {"a": 2})["result"] is False
    assert e("!a", {"a": 2})["result"] == ""
    assert e("a && true", {"a": 1})["result"] == ""                                  # && needs bools
    assert e("a > 1 ? 'big' : 'small'", {"a": 2})["result"] == "big"
    assert e("a ? 1 : 2", {"a": 1})["result"] == ""                                  # ? needs a bool
    assert e("[my var] + 1", {"my var": 1})["result"] == 2.0
    assert e("true")["result"] is True
    assert e("'it\\'s'")["result"] == "it's"
    for bad, word in [("a & 1", "'&'"), ("a | 1", "'|'"), ("a ^ 1", "'^'"), ("a << 1", "'<<'"), ("a =~ 'x'", "'=~'"), ("~a", "'~'"),
                      ("a ?? 1", "'??'"), ("a.b + 1", "accessor"), ("2 ** 3 ** 2", "chained"), ("a ? 1 : b ? 2 : 3", "nested"),
                      ("a ? 1", "'?' without ':'"), ("'2024-01-02' < a", "looks like a date"), ("foo(1)", "unknown function 'foo'"),
                      ("s2CellNeighbors(a, b)", "s2CellNeighbors"), ("geoHashWithNeighbors(a, b)", "geoHashWithNeighbors"),
                      ("1 +", "unexpected"), ("(1, 2", "expected ')'"), ("'abc", "unclosed"), ("max", "argument list"), ("1 2", "unexpected")]:
        with pytest.raises(ValueError) as ei:
            e(bad)
        assert word in str(ei.value) and "expression" in str(ei.value), (bad, str(ei.value))


def test_expr_lang_subset_semantics(H):
    e = lambda x, v=None: apply(H, "expr", x, v if v is not None else {})
    r = e("1 + 2 * 3")
    assert r["result"] == 7 and r["type"] == "int"
    assert e("7 / 2") == {"kind": "ExprNormalizer", "type": "float64", "result": 3.5}    # `/` is always float
    assert e("7 % 4")["result"] == 3 and e("-7 % 4")["result"] == -3
    assert e("7.5 % 2")["result"] == ""                                              # % is integer-only
    assert e("1 % 0")["result"] == ""
    assert e("2 ** 3 ** 2")["result"] == 512.0 and e("2 ^ 3")["result"] == 8.0       # right-associative pow, float
    assert e("-2 ** 2")["result"] == -4.0                                            # unary minus binds looser than **
    assert e("1 + 2.5")["type"] == "float64"
    assert e("1 == 1.0")["result"] is True                                           # runtime.Equal across numeric kinds
    assert e("'1' == 1")["result"] is False
    assert e("a", {})["type"] == "nil"                                               # AllowUndefinedVariables
    assert e("a ?? 'dflt'", {})["result"] == "dflt" and e("a ?? 'dflt'", {"a": "v"})["result"] == "v"
    assert e("user.age + 1", {"user": {"age": 41}})["result"] == 42
    assert e('user["age"] + 1', {"user": {"age": 41}})["result"] == 42
    assert e("user.nosuch", {"user": {}})["type"] == "nil"
    assert e("nosuch.x", {})["result"] == ""                                         # fetch from nil is an error …
    assert e("nosuch?.x", {})["type"] == "nil"                                       # … unless the chain is optional
    assert e("a in [1, 2, 3]", {"a": 2})["result"] is True and e("a not in [1, 2, 3]", {"a": 2})["result"] is False
    assert e("'k' in m", {"m": {"k": 1}})["result"] is True
    assert e("a in nosuch", {"a": 1})["result"] is False
    assert e("a > 1 and not (b > 1)", {"a": 2, "b": 0})["result"] is True
    assert e("a > 1 or nosuch.x > 1", {"a": 2})["result"] is True                    # short circuit
    assert e("not a", {"a": 1})["result"] == ""
    assert e("a contains 'ell' && a startsWith 'he' && a endsWith 'lo'", {"a": "hello"})["result"] is True
    assert e("'a' + 'b'")["result"] == "ab" and e("'a' + 1")["result"] == ""         # + joins strings only with strings
    assert e("int('42') + int(3.9)")["result"] == 45 and e("int('x')")["result"] == ""
    assert e("float(2) / 4")["result"] == 0.5 and e("string(12) + string(1.5)")["result"] == "121.5"
    assert e("len('héllo') + len([1, 2])")["result"] == 7 and e("abs(-3)")["result"] == 3 and e("abs(-3.5)")["result"] == 3.5
    assert e("a ? 1 : 2", {"a": True})["result"] == 1 and e("a ? 1 : 2", {"a": 1})["result"] == ""
    assert e("1e3 + 0x10")["result"] == 1016.0
    assert e("currentTime - item.publish_time > 86400 ? 0 : 1", {"currentTime": 1700000000, "item": {"publish_time": 1699990000}})["result"] == 1
    for bad, word in [("a | upper()", "'|'"), ("1..3", "'..'"), ("a matches 'x'", "matches"), ("{'a': 1}", "map literals"),
                      ("filter(a, # > 1)", "unknown function 'filter'"), ("a ?: 1", "elvis"), ("a.b()", "method call"), ("a[0]", "indexing"),
                      ("upper(a)", "unknown function 'upper'"), ("let x = 1; x", "'let'"), ("1_000", "digit separators"),
                      ("s2CellNeighbors(1, 2)", "s2CellNeighbors"), ("1 +", "unexpected"), ("[1, 2", "expected ']'")]:
        with pytest.raises(ValueError) as ei:
            e(bad)
        assert word in str(ei.value) and '"expr"' in str(ei.value), (bad, str(ei.value))


def test_feature_ops(H):
    """raw / compose / delete / batch_raw / new / context feature ops (op.go, delete_feature_op.go, batch_raw_feature_op.go,
    new_feature_op.go), incl. the asymmetry of the two StringProperty methods: an item's float64 prints truncated
    (item.go:114-115), a user's with strconv 'f', -1 (user.go:181-182)."""
    spec = {
        "features": [
            {"FeatureType": "raw_feature", "FeatureStore": "user", "FeatureName": "age_s", "FeatureSource": "user:age"},
            {"FeatureType": "raw_feature", "FeatureStore": "user", "FeatureName": "w_s", "FeatureSource": "user:weight", "RemoveFeatureSource": True},
            {"FeatureType": "compose_feature", "FeatureStore": "user", "FeatureName": "combo", "FeatureSource": "user:age,user:city"},
            {"FeatureType": "batch_raw_feature", "FeatureStore": "user", "FeatureName": "b1,b2", "FeatureSource": "user:age,user:city"},
            {"FeatureType": "batch_raw_feature", "FeatureStore": "user", "FeatureName": "c1", "FeatureSource": "user:age,user:city"},
            {"FeatureType": "context_feature", "FeatureStore": "user", "FeatureName": "ctx"},
            {"FeatureType": "new_feature", "FeatureStore": "user", "FeatureName": "is_adult", "Normalizer": "expression", "Expression": "age >= 18"},
            {"FeatureType": "new_feature", "FeatureStore": "user", "FeatureName": "age2", "Normalizer": "expr", "Expression": "user.age * 2"},
            {"FeatureType": "new_feature", "FeatureStore": "user", "FeatureName": "wk", "Normalizer": "weekday"},
            {"FeatureType": "new_feature", "FeatureStore": "user", "FeatureName": "no_norm", "Normalizer": "nope"},
            {"FeatureType": "delete_feature", "FeatureStore": "user", "FeatureSource": "user:city,junk"},
            {"FeatureType": "raw_feature", "FeatureStore": "item", "FeatureName": "u_age", "FeatureSource": "user:age"},
            {"FeatureType": "raw_feature", "FeatureStore": "item", "FeatureName": "price_s", "FeatureSource": "item:price"},
            {"FeatureType": "compose_feature", "FeatureStore": "item", "FeatureName": "cross", "FeatureSource": "user:age,item:id,item:cat"},
            {"FeatureType": "batch_raw_feature", "FeatureStore": "item", "FeatureName": "x1,x2", "FeatureSource": "user:age,item:cat"},
            {"FeatureType": "new_feature", "FeatureStore": "item", "FeatureName": "from_recall", "FeatureSource": "item:recall_name",
             "Normalizer": "expression", "Expression": "recall_name == 'vec'"},
            {"FeatureType": "new_feature", "FeatureStore": "item", "FeatureName": "cheap", "FeatureSource": "item:price",
             "Normalizer": "expression", "Expression": "price < 10"},
            {"FeatureType": "new_feature", "FeatureStore": "item", "FeatureName": "age_gap", "FeatureSource": "user:age",
             "Normalizer": "expression", "Expression": "age - 30"},
            {"FeatureType": "new_feature", "FeatureStore": "item", "FeatureName": "fresh", "Normalizer": "expression",
             "Expression": "currentTime - publish_time < 3600"},
            {"FeatureType": "delete_feature", "FeatureStore": "item", "FeatureSource": "item:cat"},
        ],
        "user": {"id": "u", "properties": {"age": 33, "weight": 70.5, "city": "hz", "junk": 1}},
        "items": [{"id": "i1", "retrieve_id": "vec", "properties": {"price": 9.99, "cat": "a", "publish_time": 1699999000}},
                  {"id": "i2", "retrieve_id": "hot", "properties": {"price": 25.0, "cat": "b", "publish_time": 1690000000}}],
        "context_features": {"device": "ios", "hour": 7},
        "clock_ms": 1700000000000,
    }
    out = load(H, spec)
    u = out["user"]
    assert u["age_s"] == "33" and u["w_s"] == "70.5" and "weight" not in u
    assert u["combo"] == "_33_hz" and u["b1"] == "33" and u["b2"] == "hz" and "c1" not in u
    assert u["device"] == "ios" and u["hour"] == 7
    assert u["is_adult"] == 1 and u["age2"] == 66 and "no_norm" not in u
    assert u["wk"] == time.localtime(1700000000).tm_wday
    assert "city" not in u and "junk" not in u
    i1, i2 = out["items"]
    assert i1["u_age"] == "33" and i1["price_s"] == "9" and i2["price_s"] == "25"     # item float64 → strconv.Itoa(int(v))
    assert i1["cross"] == "cross_33_i1_a" and i2["cross"] == "cross_33_i2_b"
    assert (i1["x1"], i1["x2"], i2["x2"]) == ("33", "a", "b")
    assert (i1["from_recall"], i2["from_recall"]) == (1, 0) and (i1["cheap"], i2["cheap"]) == (1, 0)
    assert i1["age_gap"] == 3.0 and type(i1["age_gap"]) is float                      # govaluate arithmetic is float64
    assert (i1["fresh"], i2["fresh"]) == (1, 0)
    assert "cat" not in i1 and "cat" not in i2
    # load-time refusals: an unknown FeatureType is the reference's panic; an expression outside the subset is refused by name
    with pytest.raises(ValueError, match="not find feature type:bogus"):
        load(H, {"features": [{"FeatureType": "bogus"}], "user": None, "items": []})
    with pytest.raises(ValueError, match="feature \"f\".*unknown function 'map'"):
        load(H, {"features": [{"FeatureType": "new_feature", "FeatureStore": "item", "FeatureName": "f", "Normalizer": "expr",
                               "Expression": "map(item.tags, # + 1)"}], "user": None, "items": []})


def test_scene_feature_configs_are_checked_when_the_engine_loads(H):
    """FeatureConfs / UserFeatureConfs (recconf.go:52-53,169-176): a FeatureType the reference panics on (op.go:32) or an expression
    outside the subset stops Engine::Create, named by key and scene — before any device is touched."""
    H.ph_engine_create.restype = C.c_void_p
    H.ph_engine_create.argtypes = [C.c_char_p]
    feat = {"FeatureType": "new_feature", "FeatureStore": "item", "FeatureName": "b", "Normalizer": "expr", "Expression": "item.tags | len()"}
    cfg = {"FeatureConfs": {"home_feed": {"FeatureLoadConfs": [{"Features": [feat]}]}}}
    assert not H.ph_engine_create(json.dumps(cfg).encode())
    assert b"FeatureConfs[home_feed]" in H.ph_last_error() and b"'|'" in H.ph_last_error()
    cfg = {"UserFeatureConfs": {"feed2": {"FeatureLoadConfs": [{"Features": [dict(feat, FeatureType="zzz")]}]}}}
    assert not H.ph_engine_create(json.dumps(cfg).encode())
    assert b"UserFeatureConfs[feed2]: not find feature type:zzz" in H.ph_last_error()
