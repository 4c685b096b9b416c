// feature.cpp — host mirror of pairec's service/feature: NewNormalizer and the FeatureOp family behind Feature.LoadFeatures
// ("simple normalizers"; normalizer.go:19-41, op.go:17-33, new_feature_op.go, feature.go:17-41).
//
// Host glue, not on the device path: user- and item-side property transforms that run before the rank call.  The reference
// delegates its two expression normalizers to third-party evaluators that are go.mod dependencies, absent from its tree:
//   "expression" → github.com/Knetic/govaluate v3.0.1-0.20171022003610-9aa49832a739 (normalizer.go:112-138)
//   "expr"       → github.com/expr-lang/expr v1.17.6, expr.AllowUndefinedVariables() (normalizer.go:140-171)
// and its geo / hash functions to golang/geo (s2), mmcloughlin/geohash v0.10.0, spaolacci/murmur3 v1.1.0 and cespare/xxhash/v2
// (utils/govaluate_functions.go).  Their published algorithms are restated here for a SUBSET, pinned by the reference's own
// tests (normalizer_test.go, feature_test.go — tests/golden/reference_known_answers.json "feature_normalizer"); everything
// outside the subset is refused BY NAME when the normalizer is built, never approximated:
//   both languages: number / string / bool literals, variables, ( ), unary - and !, + - * / %, ** (one per operand pair),
//       == != < <= > >=, in, && ||, cond ? a : b (un-nested), calls of utils.GovaluateFunctions except s2CellNeighbors and
//       geoHashWithNeighbors (their neighbour walks have no reference-side vector to pin them);
//   govaluate: every number is a float64 (its parameter sanitiser casts), `in (a, b)` lists, [bracketed names]; refused: the
//       bitwise / shift / regex operators, ??, accessors (a.b), chained **, nested ternaries, date-like string literals
//       (govaluate turns those into times);
//   expr-lang: ints stay ints (10 > 8 ? 10 : 8 is int 10), `/` is float, `%` is integer-only, `^` = `**` (right-associative),
//       member access a.b / a["b"] / a?.b, `in [a, b]` / `not in`, and / or / not, ??, contains / startsWith / endsWith, the
//       builtins int float string len abs; an undefined variable is nil; refused: pipes, lambdas and the collection builtins,
//       ranges, map literals, matches, the other builtins.
#include "pairec_host.hpp"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <memory>

namespace pairec {
namespace feature {

using V = json::Value;

// ---- values ------------------------------------------------------------------------------------------------
static V Nil() { return V(); }
static V Bool(bool b) { V v; v.type = V::Bool; v.b = b; return v; }
static V Int(long long i) { V v; v.type = V::Number; v.is_int = true; v.i = i; v.num = (double)i; return v; }
static V U64(uint64_t u) { V v = Int((long long)u); v.is_u64 = true; v.num = (double)u; return v; }
static V Float(double d) { V v; v.type = V::Number; v.num = d; return v; }
static V Str(std::string s) { return V::Str(std::move(s)); }
static V List(std::vector<V> l) { V v; v.type = V::Array; v.arr = std::move(l); return v; }
static bool is_num(const V& v) { return v.type == V::Number; }
static bool is_str(const V& v) { return v.type == V::String; }

// strconv.FormatFloat(x, 'f', -1, 64): the shortest digits that round-trip, never in exponent form
static std::string go_format_f(double x) {
    if (x != x) return "NaN";
    if (std::isinf(x)) return x > 0 ? "+Inf" : "-Inf";
    if (x == 0) return std::signbit(x) ? "-0" : "0";
    char buf[40];
    for (int prec = 1; prec <= 17; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, x);
        if (strtod(buf, nullptr) == x) break;
    }
    std::string m(buf);
    const size_t epos = m.find('e');
    const int dexp = atoi(m.c_str() + epos + 1);
    std::string digits;
    for (size_t i = 0; i < epos; ++i)
        if (m[i] >= '0' && m[i] <= '9') digits.push_back(m[i]);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const std::string sign = x < 0 ? "-" : "";
    if (dexp >= 0) {
        if ((int)digits.size() <= dexp + 1) return sign + digits + std::string((size_t)(dexp + 1) - digits.size(), '0');
        return sign + digits.substr(0, (size_t)dexp + 1) + "." + digits.substr((size_t)dexp + 1);
    }
    return sign + "0." + std::string((size_t)(-dexp - 1), '0') + digits;
}

// Go's int(float64) on amd64 (cvttsd2si): truncation; NaN and values outside int64 give the "integer indefinite" 0x8000000000000000
static long long f2i(double d) {
    if (!(d > -9223372036854775808.0 && d < 9223372036854775808.0)) return (long long)0x8000000000000000ull;
    return (long long)d;
}

// utils.ToString (utils/type.go:120-141)
static std::string to_string(const V& v, const std::string& def) {
    if (v.type == V::String) return v.str;
    if (v.type == V::Number) {
        if (v.is_u64) return def;                              // uint64 is not in its switch
        return v.is_int ? std::to_string(v.i) : go_format_f(v.num);
    }
    return def;
}
// utils.ToInt (utils/type.go:11-42)
static long long to_int(const V& v, long long def) {
    if (v.type == V::Number) return v.is_int ? v.i : f2i(v.num);
    if (v.type == V::String) {
        char* e = nullptr;
        const long long r = strtoll(v.str.c_str(), &e, 10);
        if (!v.str.empty() && e && *e == '\0' && v.str[0] != ' ') return r;
        return def;
    }
    return def;
}
// utils.ToFloatArray on what JSON can carry: a list of numbers (strings parse, as ToFloat does)
static std::vector<double> to_float_array(const V& v) {
    std::vector<double> out;
    if (v.type == V::Array)
        for (const V& e : v.arr) out.push_back(ToFloat(e, 0.0));
    return out;
}

// fmt %v of a value (govaluate's string concatenation: fmt.Sprintf("%v%v", left, right))
static std::string sprint_v(const V& v) {
    switch (v.type) {
        case V::String: return v.str;
        case V::Bool: return v.b ? "true" : "false";
        case V::Number:
            if (v.is_u64) return std::to_string((unsigned long long)v.i);
            return v.is_int ? std::to_string(v.i) : GoFmtFloat(v.num);
        case V::Array: {
            std::string s = "[";
            for (size_t i = 0; i < v.arr.size(); ++i) s += (i ? " " : "") + sprint_v(v.arr[i]);
            return s + "]";
        }
        default: return "<nil>";
    }
}

// ---- clock -------------------------------------------------------------------------------------------------
static std::atomic<long long> g_clock_ms{0};
void SetClockForTest(long long unix_millis) { g_clock_ms = unix_millis; }
static long long now_ms() {
    const long long c = g_clock_ms;
    if (c) return c;
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}
static tm local_now() {
    const time_t t = (time_t)(now_ms() / 1000);
    tm r;
    localtime_r(&t, &r);
    return r;
}
// the week of Time.ISOWeek (ISO 8601: weeks start on Monday, week 1 holds the year's first Thursday)
static int iso_week(const tm& t) {
    const int wday = t.tm_wday == 0 ? 7 : t.tm_wday;           // Monday 1 … Sunday 7
    const int yday = t.tm_yday + 1;
    int week = (yday - wday + 10) / 7;
    const int year = t.tm_year + 1900;
    auto weeks_in = [](int y) {                                 // 53 when Jan 1 is a Thursday, or a Wednesday of a leap year
        const int p = (y + y / 4 - y / 100 + y / 400) % 7, q = ((y - 1) + (y - 1) / 4 - (y - 1) / 100 + (y - 1) / 400) % 7;
        return (p == 4 || q == 3) ? 53 : 52;
    };
    if (week < 1) return weeks_in(year - 1);
    if (week > weeks_in(year)) return 1;
    return week;
}

// ---- hashes ------------------------------------------------------------------------------------------------
// spaolacci/murmur3 Sum32: MurmurHash3_x86_32, seed 0
static uint32_t murmur3_32(const std::string& s) {
    const uint8_t* p = (const uint8_t*)s.data();
    const size_t n = s.size();
    uint32_t h = 0;
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    auto rotl = [](uint32_t x, int r) { return (x << r) | (x >> (32 - r)); };
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        uint32_t k;
        memcpy(&k, p + i, 4);
        k *= c1; k = rotl(k, 15); k *= c2;
        h ^= k; h = rotl(h, 13); h = h * 5 + 0xe6546b64u;
    }
    uint32_t k = 0;
    switch (n & 3) {
        case 3: k ^= (uint32_t)p[i + 2] << 16; /* fallthrough */
        case 2: k ^= (uint32_t)p[i + 1] << 8;  /* fallthrough */
        case 1: k ^= p[i]; k *= c1; k = rotl(k, 15); k *= c2; h ^= k;
    }
    h ^= (uint32_t)n;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// cespare/xxhash/v2 Sum64String: XXH64, seed 0
static uint64_t xxhash64(const std::string& s) {
    const uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull,
                   P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto rd64 = [](const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; };
    auto rd32 = [](const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; };
    auto merge = [&](uint64_t acc, uint64_t val) { acc ^= round(0, val); return acc * P1 + P4; };
    const uint8_t* p = (const uint8_t*)s.data();
    const uint8_t* const end = p + s.size();
    uint64_t h;
    if (s.size() >= 32) {
        uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
        for (; p + 32 <= end; p += 32) {
            v1 = round(v1, rd64(p)); v2 = round(v2, rd64(p + 8)); v3 = round(v3, rd64(p + 16)); v4 = round(v4, rd64(p + 24));
        }
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = P5;
    }
    h += (uint64_t)s.size();
    for (; p + 8 <= end; p += 8) { h ^= round(0, rd64(p)); h = rotl(h, 27) * P1 + P4; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
    for (; p < end; ++p) { h ^= (uint64_t)*p * P5; h = rotl(h, 11) * P1; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// ---- geo ---------------------------------------------------------------------------------------------------
// mmcloughlin/geohash EncodeWithPrecision: each coordinate scaled to 32 bits (uint32((x + r) / 2r * 2^32)), longitude on the odd
// bits, the top 5 * chars bits in its base-32 alphabet.  Coordinates on the upper edge (lat 90, lng 180) overflow the uint32
// conversion in Go (implementation-defined): refused here.
static bool geohash_encode(double lat, double lng, long long chars, std::string* out, std::string* err) {
    if (chars < 1 || chars > 12) { *err = "geoHash: precision must be 1..12"; return false; }
    if (!(lat >= -90.0 && lat < 90.0 && lng >= -180.0 && lng < 180.0)) { *err = "geoHash: lat must be in [-90, 90), lng in [-180, 180)"; return false; }
    auto enc_range = [](double x, double r) { return (uint32_t)((x + r) / (2 * r) * 4294967296.0); };
    auto spread = [](uint32_t x) {
        uint64_t X = x;
        X = (X | (X << 16)) & 0x0000ffff0000ffffull;
        X = (X | (X << 8)) & 0x00ff00ff00ff00ffull;
        X = (X | (X << 4)) & 0x0f0f0f0f0f0f0f0full;
        X = (X | (X << 2)) & 0x3333333333333333ull;
        X = (X | (X << 1)) & 0x5555555555555555ull;
        return X;
    };
    const uint64_t full = spread(enc_range(lat, 90.0)) | (spread(enc_range(lng, 180.0)) << 1);
    uint64_t h = full >> (64 - 5 * chars);
    static const char* alphabet = "0123456789bcdefghjkmnpqrstuvwxyz";
    std::string s((size_t)chars, '0');
    for (long long i = chars - 1; i >= 0; --i) { s[(size_t)i] = alphabet[h & 31]; h >>= 5; }
    *out = s;
    return true;
}

// golang/geo s2.CellIDFromLatLng(ll).Parent(level): unit vector → cube face → (u, v) → quadratic (s, t) → 30-bit (i, j) →
// position along the Hilbert curve of the face (lookup by 4 bits of i and j per step), then the level's parent
static uint64_t s2_cell_id(double lat_deg, double lng_deg, int level) {
    static int lookup_pos[1024];
    static const bool init = [] {
        static const int pos_to_ij[4][4] = {{0, 1, 3, 2}, {0, 2, 3, 1}, {3, 2, 0, 1}, {3, 1, 0, 2}};
        static const int pos_to_orientation[4] = {1, 0, 0, 3};         // swap, 0, 0, invert | swap
        struct R {
            static void cell(int level, int i, int j, int orig, int pos, int orientation) {
                if (level == 4) { lookup_pos[(((i << 4) + j) << 2) + orig] = (pos << 2) + orientation; return; }
                ++level; i <<= 1; j <<= 1; pos <<= 2;
                const int* r = pos_to_ij[orientation];
                for (int k = 0; k < 4; ++k) cell(level, i + (r[k] >> 1), j + (r[k] & 1), orig, pos + k, orientation ^ pos_to_orientation[k]);
            }
        };
        for (int o = 0; o < 4; ++o) R::cell(0, 0, 0, o, 0, o);
        return true;
    }();
    (void)init;
    const double phi = lat_deg * (M_PI / 180.0), theta = lng_deg * (M_PI / 180.0);
    const double cosphi = std::cos(phi);
    const double p[3] = {std::cos(theta) * cosphi, std::sin(theta) * cosphi, std::sin(phi)};
    int axis = 0;
    if (std::fabs(p[1]) > std::fabs(p[axis])) axis = 1;
    if (std::fabs(p[2]) > std::fabs(p[axis])) axis = 2;
    const int face = axis + (p[axis] < 0 ? 3 : 0);
    const double x = p[0], y = p[1], z = p[2];
    double u, v;
    switch (face) {
        case 0: u = y / x; v = z / x; break;
        case 1: u = -x / y; v = z / y; break;
        case 2: u = -x / z; v = -y / z; break;
        case 3: u = z / x; v = y / x; break;
        case 4: u = z / y; v = -x / y; break;
        default: u = -y / z; v = -x / z;
    }
    auto uv_to_st = [](double a) { return a >= 0 ? 0.5 * std::sqrt(1 + 3 * a) : 1 - 0.5 * std::sqrt(1 - 3 * a); };
    auto st_to_ij = [](double s) {
        const long long m = 1ll << 30;
        long long r = (long long)std::floor((double)m * s);
        return (int)(r < 0 ? 0 : (r > m - 1 ? m - 1 : r));
    };
    const int i = st_to_ij(uv_to_st(u)), j = st_to_ij(uv_to_st(v));
    uint64_t n = (uint64_t)face << 60;
    int bits = face & 1;
    for (int k = 7; k >= 0; --k) {
        bits += ((i >> (k * 4)) & 15) << 6;
        bits += ((j >> (k * 4)) & 15) << 2;
        bits = lookup_pos[bits];
        n |= (uint64_t)(bits >> 2) << (k * 8);
        bits &= 3;
    }
    const uint64_t leaf = n * 2 + 1;
    const uint64_t lsb = 1ull << (2 * (30 - level));
    return (leaf & (0 - lsb)) | lsb;
}

// ---- utils.GovaluateFunctions --------------------------------------------------------------------------------
static const char* const kFunctions[] = {"getString", "trim", "trimPrefix", "replace", "round", "hash", "hash32", "toFloat64",
                                         "log", "log10", "log2", "max", "min", "pow", "s2CellID", "geoHash", "haversine",
                                         "sphereDistance", "timestamp", "maxIndex", "maxValue"};
static const char* const kRefusedFunctions[] = {"s2CellNeighbors", "geoHashWithNeighbors"};
bool HasFunction(const std::string& name) {
    for (const char* f : kFunctions) if (name == f) return true;
    return false;
}
static std::vector<uint32_t> runes(const std::string& s) {          // UTF-8 → code points (invalid bytes as themselves)
    std::vector<uint32_t> r;
    for (size_t i = 0; i < s.size();) {
        const unsigned char c = (unsigned char)s[i];
        int n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1;
        if (i + (size_t)n > s.size()) n = 1;
        uint32_t cp = n == 1 ? c : (c & (0xff >> (n + 1)));
        for (int k = 1; k < n; ++k) cp = (cp << 6) | ((unsigned char)s[i + (size_t)k] & 0x3f);
        r.push_back(cp | ((uint32_t)n << 24));                      // the byte length rides in the top byte
        i += (size_t)n;
    }
    return r;
}
static std::string go_trim(const std::string& s, const std::string& cutset) {     // strings.Trim
    const auto rs = runes(s), cs = runes(cutset);
    auto in_set = [&](uint32_t r) { for (uint32_t c : cs) if (c == r) return true; return false; };
    size_t b = 0, e = rs.size(), off_b = 0, off_e = s.size();
    while (b < e && in_set(rs[b])) { off_b += rs[b] >> 24; ++b; }
    while (e > b && in_set(rs[e - 1])) { off_e -= rs[e - 1] >> 24; --e; }
    return s.substr(off_b, off_e - off_b);
}
static std::string go_replace_all(const std::string& s, const std::string& from, const std::string& to) {
    std::string out;
    if (from.empty()) {                                             // strings.ReplaceAll: `to` before every rune and at the end
        size_t off = 0;
        for (uint32_t r : runes(s)) { out += to; out += s.substr(off, r >> 24); off += r >> 24; }
        return out + to;
    }
    size_t pos = 0;
    for (;;) {
        const size_t hit = s.find(from, pos);
        if (hit == std::string::npos) break;
        out += s.substr(pos, hit - pos) + to;
        pos = hit + from.size();
    }
    return out + s.substr(pos);
}

bool CallFunction(const std::string& name, const std::vector<V>& a, V* out, std::string* err) {
    auto fail = [&](const char* m) { *err = name + ": " + m; return false; };
    auto f = [&](size_t i) { return ToFloat(a[i], 0.0); };
    const size_t n = a.size();
    if (name == "getString") {                                      // govaluate_functions.go:21-32
        if (n == 0) return fail("args should not empty");
        if (!(is_str(a[0]) && a[0].str.empty())) { *out = a[0]; return true; }
        *out = n > 1 ? a[1] : Str("");
        return true;
    }
    if (name == "trim") {                                           // :33-42
        if (n != 2) return fail("args length not equal 2");
        *out = Str(go_trim(to_string(a[0], ""), to_string(a[1], "")));
        return true;
    }
    if (name == "trimPrefix") {                                     // :43-51
        if (n != 2) return fail("args length not equal 2");
        const std::string s = to_string(a[0], ""), p = to_string(a[1], "");
        *out = Str(s.compare(0, p.size(), p) == 0 ? s.substr(p.size()) : s);
        return true;
    }
    if (name == "replace") {                                        // :52-61
        if (n != 3) return fail("args length not equal 3");
        *out = Str(go_replace_all(to_string(a[0], ""), to_string(a[1], ""), to_string(a[2], "")));
        return true;
    }
    if (name == "round") {                                          // :62-74: math.Round, or truncation at n decimals
        if (n == 1) { *out = Float(std::round(f(0))); return true; }
        if (n == 2) { const double m = std::pow(10.0, f(1)); *out = Float(std::trunc(f(0) * m) / m); return true; }
        return fail("wrong number of arguments");
    }
    if (name == "hash") {                                           // :75-81
        if (n != 1) return fail("args length not equal 1");
        *out = U64(xxhash64(to_string(a[0], "")));
        return true;
    }
    if (name == "hash32") {                                         // :82-88
        if (n != 1) return fail("args length not equal 1");
        *out = Float((double)murmur3_32(to_string(a[0], "")));
        return true;
    }
    if (name == "toFloat64" || name == "log" || name == "log10" || name == "log2") {   // :89-112
        if (n != 1) return fail("args length not equal 1");
        const double x = f(0);
        *out = Float(name == "toFloat64" ? x : name == "log" ? std::log(x) : name == "log10" ? std::log10(x) : std::log2(x));
        return true;
    }
    if (name == "max" || name == "min" || name == "pow") {          // :113-130 (math.Max / math.Min: NaN and signed-zero rules)
        if (n != 2) return fail("args length not equal 2");
        const double x = f(0), y = f(1);
        if (name == "pow") { *out = Float(std::pow(x, y)); return true; }
        double r;
        if (name == "max") {
            if (std::isinf(x) && x > 0) r = x; else if (std::isinf(y) && y > 0) r = y;
            else if (x != x || y != y) r = NAN;
            else if (x == 0 && y == 0) r = std::signbit(x) ? y : x;
            else r = x > y ? x : y;
        } else {
            if (std::isinf(x) && x < 0) r = x; else if (std::isinf(y) && y < 0) r = y;
            else if (x != x || y != y) r = NAN;
            else if (x == 0 && y == 0) r = std::signbit(x) ? x : y;
            else r = x < y ? x : y;
        }
        *out = Float(r);
        return true;
    }
    if (name == "s2CellID") {                                       // :131-147
        if (n < 2) return fail("args must have lat and lng params");
        const long long level = n > 2 ? to_int(a[2], 15) : 15;
        if (level < 0 || level > 30) return fail("level must be 0..30");
        if (!std::isfinite(f(0)) || !std::isfinite(f(1))) return fail("lat / lng must be finite");
        *out = Int((long long)s2_cell_id(f(0), f(1), (int)level));
        return true;
    }
    if (name == "geoHash") {                                        // :173-184
        if (n < 2) return fail("args must have lat and lng params");
        std::string h, e;
        if (!geohash_encode(f(0), f(1), n > 2 ? to_int(a[2], 6) : 6, &h, &e)) { *err = e; return false; }
        *out = Str(h);
        return true;
    }
    if (name == "haversine") {                                      // :204-228 (lng1, lat1, lng2, lat2) → km
        if (n != 4) return fail("args length not equal 4");
        auto rad = [](double d) { return d * M_PI / 180; };
        const double la1 = rad(f(1)), la2 = rad(f(3)), lo1 = rad(f(0)), lo2 = rad(f(2));
        const double dla = la2 - la1, dlo = lo2 - lo1;
        const double h = std::sin(dla / 2) * std::sin(dla / 2) + std::cos(la1) * std::cos(la2) * std::sin(dlo / 2) * std::sin(dlo / 2);
        *out = Float(6371.0 * (2 * std::atan2(std::sqrt(h), std::sqrt(1 - h))));
        return true;
    }
    if (name == "sphereDistance") {                                 // :229-256
        if (n != 4) return fail("args length not equal 4");
        auto rad = [](double d) { return d * M_PI / 180; };
        const double la1 = rad(f(1)), la2 = rad(f(3)), dlo = rad(f(2) - f(0));
        double c = std::sin(la1) * std::sin(la2) + std::cos(la1) * std::cos(la2) * std::cos(dlo);
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        *out = Float(std::acos(c) * 6371.0);
        return true;
    }
    if (name == "timestamp") {                                      // :258-267
        std::string unit = n > 0 ? to_string(a[0], "") : "";
        for (char& c : unit) c = (char)tolower((unsigned char)c);
        const long long ms = now_ms();
        *out = Float(unit == "ms" || unit == "millisecond" || unit == "milliseconds" ? (double)ms : (double)(ms / 1000));
        return true;
    }
    if (name == "maxIndex" || name == "maxValue") {                 // :268-305 + findMax :311-330: the first maximum
        if (n != 1) return fail("expects exactly one argument");
        const std::vector<double> xs = to_float_array(a[0]);
        if (xs.empty()) return fail("argument must not be empty");
        size_t best = 0;
        for (size_t i = 1; i < xs.size(); ++i) if (xs[i] > xs[best]) best = i;
        *out = name == "maxIndex" ? Int((long long)best) : Float(xs[best]);
        return true;
    }
    *err = "unknown function '" + name + "'";
    return false;
}

// ---- the two expression languages ----------------------------------------------------------------------------
namespace {
enum class Dialect { Govaluate, ExprLang };

struct Tok {
    enum K { End, Num, Str, Ident, Op, LParen, RParen, LBrack, RBrack, Comma } k = End;
    std::string s;
    V num;
    bool bracket_name = false;      // govaluate's [name with anything]
};

struct Node {
    enum K { Lit, Var, Member, Unary, Binary, Ternary, Call, ListLit } k = Lit;
    std::string op;                 // operator, variable / member / function name
    V lit;
    bool optional = false;          // a?.b
    std::vector<std::unique_ptr<Node>> kids;
};
using NodeP = std::unique_ptr<Node>;

static bool looks_like_a_date(const std::string& s) {
    auto dig = [&](size_t i) { return i < s.size() && s[i] >= '0' && s[i] <= '9'; };
    if (dig(0) && dig(1) && dig(2) && dig(3) && s.size() > 9 && s[4] == '-' && dig(5) && dig(6) && s[7] == '-' && dig(8) && dig(9)) return true;
    static const char* const names[] = {"Mon", "Tue", "Wed", "Thu", "Fri", "Sat", "Sun", "Monday", "Tuesday", "Wednesday", "Thursday",
                                        "Friday", "Saturday", "Sunday"};
    for (const char* d : names) {
        const size_t n = strlen(d);
        if (s.compare(0, n, d) == 0 && s.size() > n && (s[n] == ' ' || s[n] == ',')) return true;
    }
    const size_t colon = s.find(':');                              // time.Kitchen "3:04PM"
    if (colon != std::string::npos && colon >= 1 && colon <= 2 && dig(0) && s.size() == colon + 5 && dig(colon + 1) && dig(colon + 2) &&
        (s.compare(colon + 3, 2, "AM") == 0 || s.compare(colon + 3, 2, "PM") == 0)) return true;
    return false;
}

class Lexer {
public:
    Lexer(const std::string& src, Dialect d) : s_(src), d_(d) {}
    bool Run(std::vector<Tok>* out, std::string* err) {
        size_t p = 0;
        const size_t n = s_.size();
        while (p < n) {
            const char c = s_[p];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { ++p; continue; }
            Tok t;
            if ((c >= '0' && c <= '9') || (c == '.' && p + 1 < n && s_[p + 1] >= '0' && s_[p + 1] <= '9')) {
                size_t q = p;
                bool is_float = false;
                if (c == '0' && q + 1 < n && (s_[q + 1] == 'x' || s_[q + 1] == 'X')) {
                    q += 2;
                    const size_t h0 = q;
                    while (q < n && isxdigit((unsigned char)s_[q])) ++q;
                    if (q == h0) { *err = "bad hexadecimal literal"; return false; }
                    const unsigned long long hv = strtoull(s_.substr(h0, q - h0).c_str(), nullptr, 16);
                    t.k = Tok::Num;
                    t.num = d_ == Dialect::Govaluate ? Float((double)hv) : Int((long long)hv);
                    p = q;
                    out->push_back(t);
                    continue;
                }
                while (q < n && s_[q] >= '0' && s_[q] <= '9') ++q;
                // "1..3" is expr-lang's range, not a fraction
                if (q < n && s_[q] == '.' && !(q + 1 < n && s_[q + 1] == '.')) { is_float = true; ++q; while (q < n && s_[q] >= '0' && s_[q] <= '9') ++q; }
                if (d_ == Dialect::ExprLang && q < n && (s_[q] == 'e' || s_[q] == 'E')) {
                    size_t r = q + 1;
                    if (r < n && (s_[r] == '+' || s_[r] == '-')) ++r;
                    if (r < n && s_[r] >= '0' && s_[r] <= '9') { is_float = true; q = r; while (q < n && s_[q] >= '0' && s_[q] <= '9') ++q; }
                }
                if (q < n && s_[q] == '_') { *err = "digit separators ('_') are outside the supported subset"; return false; }
                const std::string lit = s_.substr(p, q - p);
                t.k = Tok::Num;
                if (d_ == Dialect::Govaluate || is_float) t.num = Float(strtod(lit.c_str(), nullptr));
                else t.num = Int(strtoll(lit.c_str(), nullptr, 10));
                p = q;
                out->push_back(t);
                continue;
            }
            if (c == '\'' || c == '"' || (c == '`' && d_ == Dialect::ExprLang)) {
                size_t q = p + 1;
                std::string lit;
                bool closed = false;
                while (q < n) {
                    const char ch = s_[q++];
                    if (ch == c) { closed = true; break; }
                    if (ch == '\\' && c != '`' && q < n) {
                        const char e = s_[q++];
                        if (d_ == Dialect::Govaluate) { lit.push_back(e); continue; }   // govaluate: the next character, literally
                        switch (e) {
                            case 'n': lit.push_back('\n'); break;
                            case 't': lit.push_back('\t'); break;
                            case 'r': lit.push_back('\r'); break;
                            case '\\': case '\'': case '"': lit.push_back(e); break;
                            default: *err = std::string("string escape '\\") + e + "' is outside the supported subset"; return false;
                        }
                        continue;
                    }
                    lit.push_back(ch);
                }
                if (!closed) { *err = "unclosed string literal"; return false; }
                if (d_ == Dialect::Govaluate && looks_like_a_date(lit)) {
                    *err = "string literal '" + lit + "' looks like a date: govaluate turns those into times, which is outside the supported subset";
                    return false;
                }
                t.k = Tok::Str;
                t.s = lit;
                p = q;
                out->push_back(t);
                continue;
            }
            if (c == '[' && d_ == Dialect::Govaluate) {               // [a variable name]
                const size_t q = s_.find(']', p);
                if (q == std::string::npos) { *err = "unclosed parameter bracket"; return false; }
                t.k = Tok::Ident;
                t.s = s_.substr(p + 1, q - p - 1);
                t.bracket_name = true;
                p = q + 1;
                out->push_back(t);
                continue;
            }
            if (isalpha((unsigned char)c) || c == '_' || (unsigned char)c >= 0x80) {
                size_t q = p;
                while (q < n && (isalnum((unsigned char)s_[q]) || s_[q] == '_' || (unsigned char)s_[q] >= 0x80 ||
                                 (d_ == Dialect::Govaluate && s_[q] == '.')))
                    ++q;
                t.k = Tok::Ident;
                t.s = s_.substr(p, q - p);
                if (d_ == Dialect::Govaluate && t.s.find('.') != std::string::npos) {
                    *err = "accessor '" + t.s + "' (a.b) is outside the supported subset";
                    return false;
                }
                p = q;
                out->push_back(t);
                continue;
            }
            static const char* const ops3[] = {"**", "==", "!=", ">=", "<=", "&&", "||", "??", "?.", "<<", ">>", "=~", "!~", ".."};
            bool matched = false;
            for (const char* o : ops3)
                if (s_.compare(p, 2, o) == 0) {
                    if (std::string(o) == "?." && !(d_ == Dialect::ExprLang && p + 2 < n && !(s_[p + 2] >= '0' && s_[p + 2] <= '9'))) continue;
                    t.k = Tok::Op; t.s = o; p += 2; matched = true;
                    break;
                }
            if (!matched) {
                switch (c) {
                    case '(': t.k = Tok::LParen; break;
                    case ')': t.k = Tok::RParen; break;
                    case '[': t.k = Tok::LBrack; break;
                    case ']': t.k = Tok::RBrack; break;
                    case ',': t.k = Tok::Comma; break;
                    case '+': case '-': case '*': case '/': case '%': case '^': case '&': case '|': case '~': case '>': case '<':
                    case '!': case '?': case ':': case '.': case '{': case '}': case '#': case ';': case '=':
                        t.k = Tok::Op; t.s = std::string(1, c); break;
                    default: *err = std::string("unexpected character '") + c + "'"; return false;
                }
                ++p;
            }
            out->push_back(t);
        }
        out->push_back(Tok());
        return true;
    }
private:
    const std::string& s_;
    Dialect d_;
};

class Parser {
public:
    Parser(std::vector<Tok> toks, Dialect d) : t_(std::move(toks)), d_(d) {}
    NodeP Parse(std::string* err) {
        NodeP n = d_ == Dialect::Govaluate ? gv_ternary() : ex_expr(0, true);
        if (n && cur().k != Tok::End) fail("unexpected '" + show(cur()) + "'");
        if (!err_.empty()) { *err = err_; return nullptr; }
        return n;
    }
private:
    std::vector<Tok> t_;
    Dialect d_;
    size_t p_ = 0;
    std::string err_;
    int depth_ = 0;

    const Tok& cur() const { return t_[p_]; }
    const Tok& peek(size_t k = 1) const { return t_[std::min(p_ + k, t_.size() - 1)]; }
    void adv() { if (p_ + 1 < t_.size()) ++p_; }
    bool is_op(const char* o) const { return cur().k == Tok::Op && cur().s == o; }
    bool is_word(const char* w) const { return cur().k == Tok::Ident && !cur().bracket_name && cur().s == w; }
    NodeP fail(const std::string& m) { if (err_.empty()) err_ = m; return nullptr; }
    static std::string show(const Tok& t) {
        switch (t.k) {
            case Tok::End: return "end of expression";
            case Tok::LParen: return "(";
            case Tok::RParen: return ")";
            case Tok::LBrack: return "[";
            case Tok::RBrack: return "]";
            case Tok::Comma: return ",";
            case Tok::Num: return sprint_v(t.num);
            default: return t.s;
        }
    }
    static NodeP mk(Node::K k, std::string op = "") { NodeP n(new Node); n->k = k; n->op = std::move(op); return n; }
    static NodeP bin(const std::string& op, NodeP a, NodeP b) {
        NodeP n = mk(Node::Binary, op);
        n->kids.push_back(std::move(a));
        n->kids.push_back(std::move(b));
        return n;
    }
    struct Depth {
        int& d;
        explicit Depth(int& x) : d(x) { ++d; }
        ~Depth() { --d; }
    };
    bool too_deep() { if (depth_ > 200) { fail("expression nests too deeply"); return true; } return false; }

    NodeP refuse_op() { return fail("operator '" + cur().s + "' is outside the supported subset"); }

    // ---------------- govaluate: separator < ternary < || < && < comparators < (bitwise, shift) < + - < * / % < ** < prefix < value
    NodeP gv_ternary() {
        Depth dd(depth_);
        if (too_deep()) return nullptr;
        NodeP c = gv_or();
        if (!c) return nullptr;
        if (is_op("??")) return refuse_op();
        if (!is_op("?")) {
            if (is_op(":")) return fail("':' without '?'");
            return c;
        }
        adv();
        NodeP a = gv_or();
        if (!a) return nullptr;
        if (is_op("?")) return fail("nested ternaries need parentheses in the supported subset");
        if (!is_op(":")) return fail("'?' without ':' is outside the supported subset");
        adv();
        NodeP b = gv_or();
        if (!b) return nullptr;
        if (is_op("?") || is_op(":")) return fail("nested ternaries need parentheses in the supported subset");
        NodeP n = mk(Node::Ternary);
        n->kids.push_back(std::move(c));
        n->kids.push_back(std::move(a));
        n->kids.push_back(std::move(b));
        return n;
    }
    NodeP gv_or() {
        NodeP l = gv_and();
        while (l && is_op("||")) { adv(); NodeP r = gv_and(); if (!r) return nullptr; l = bin("||", std::move(l), std::move(r)); }
        return l;
    }
    NodeP gv_and() {
        NodeP l = gv_cmp();
        while (l && is_op("&&")) { adv(); NodeP r = gv_cmp(); if (!r) return nullptr; l = bin("&&", std::move(l), std::move(r)); }
        return l;
    }
    NodeP gv_cmp() {
        NodeP l = gv_add();
        for (;;) {
            if (!l) return nullptr;
            std::string op;
            if (cur().k == Tok::Op && (cur().s == "==" || cur().s == "!=" || cur().s == ">" || cur().s == ">=" || cur().s == "<" || cur().s == "<=")) op = cur().s;
            else if (cur().k == Tok::Ident && !cur().bracket_name && (cur().s == "in" || cur().s == "IN")) op = "in";
            else if (cur().k == Tok::Op && (cur().s == "=~" || cur().s == "!~" || cur().s == "&" || cur().s == "|" || cur().s == "^" ||
                                            cur().s == "<<" || cur().s == ">>"))
                return refuse_op();
            else return l;
            adv();
            NodeP r = gv_add();
            if (!r) return nullptr;
            l = bin(op, std::move(l), std::move(r));
        }
    }
    NodeP gv_add() {
        NodeP l = gv_mul();
        while (l && (is_op("+") || is_op("-"))) { const std::string op = cur().s; adv(); NodeP r = gv_mul(); if (!r) return nullptr; l = bin(op, std::move(l), std::move(r)); }
        return l;
    }
    NodeP gv_mul() {
        NodeP l = gv_exp();
        while (l && (is_op("*") || is_op("/") || is_op("%"))) { const std::string op = cur().s; adv(); NodeP r = gv_exp(); if (!r) return nullptr; l = bin(op, std::move(l), std::move(r)); }
        return l;
    }
    NodeP gv_exp() {
        NodeP l = gv_prefix();
        if (l && is_op("**")) {
            adv();
            NodeP r = gv_prefix();
            if (!r) return nullptr;
            if (is_op("**")) return fail("chained '**' is outside the supported subset (write the parentheses)");
            l = bin("**", std::move(l), std::move(r));
        }
        return l;
    }
    NodeP gv_prefix() {
        Depth dd(depth_);
        if (too_deep()) return nullptr;
        if (is_op("-") || is_op("!")) {
            const std::string op = cur().s;
            adv();
            NodeP v = gv_prefix();
            if (!v) return nullptr;
            NodeP n = mk(Node::Unary, op);
            n->kids.push_back(std::move(v));
            return n;
        }
        if (is_op("~")) return refuse_op();
        return gv_value();
    }
    NodeP gv_value() {
        const Tok t = cur();
        switch (t.k) {
            case Tok::Num: { adv(); NodeP n = mk(Node::Lit); n->lit = t.num; return n; }
            case Tok::Str: { adv(); NodeP n = mk(Node::Lit); n->lit = Str(t.s); return n; }
            case Tok::LParen: {
                adv();
                if (cur().k == Tok::RParen) return fail("empty parentheses");
                std::vector<NodeP> items;
                for (;;) {
                    NodeP e = gv_ternary();
                    if (!e) return nullptr;
                    items.push_back(std::move(e));
                    if (cur().k == Tok::Comma) { adv(); continue; }
                    break;
                }
                if (cur().k != Tok::RParen) return fail("expected ')' before '" + show(cur()) + "'");
                adv();
                if (items.size() == 1) return std::move(items[0]);
                NodeP n = mk(Node::ListLit);                        // the separator operator: (a, b, c) is an array
                n->kids = std::move(items);
                return n;
            }
            case Tok::Ident: {
                adv();
                if (!t.bracket_name && (t.s == "true" || t.s == "false")) { NodeP n = mk(Node::Lit); n->lit = Bool(t.s == "true"); return n; }
                if (!t.bracket_name && cur().k == Tok::LParen) return call(t.s);
                if (!t.bracket_name && HasFunction(t.s)) return fail("function '" + t.s + "' needs its argument list");
                return mk(Node::Var, t.s);
            }
            default: return fail("unexpected '" + show(t) + "'");
        }
    }
    NodeP call(const std::string& name) {
        for (const char* r : kRefusedFunctions)
            if (name == r) return fail("function '" + name + "' is outside the supported subset (no reference vector pins its neighbour walk)");
        const bool builtin = d_ == Dialect::ExprLang && (name == "int" || name == "float" || name == "string" || name == "len" || name == "abs");
        if (!HasFunction(name) && !builtin) return fail("unknown function '" + name + "'");
        adv();                                                       // (
        NodeP n = mk(Node::Call, name);
        if (cur().k == Tok::RParen) { adv(); return n; }
        for (;;) {
            NodeP e = d_ == Dialect::Govaluate ? gv_ternary() : ex_expr(0, true);
            if (!e) return nullptr;
            n->kids.push_back(std::move(e));
            if (cur().k == Tok::Comma) { adv(); continue; }
            break;
        }
        if (cur().k != Tok::RParen) return fail("expected ')' before '" + show(cur()) + "'");
        adv();
        return n;
    }

    // ---------------- expr-lang: precedence climbing over parser/operator's table
    struct OpInfo { int prec; bool right; };
    bool ex_binary(std::string* op, OpInfo* info) {
        const Tok& t = cur();
        std::string s;
        if (t.k == Tok::Op) s = t.s;
        else if (t.k == Tok::Ident && !t.bracket_name) s = t.s;
        else return false;
        if (s == "not" && peek().k == Tok::Ident && peek().s == "in") s = "not in";
        static const struct { const char* op; int prec; bool right; } table[] = {
            {"or", 10, false}, {"||", 10, false}, {"and", 15, false}, {"&&", 15, false},
            {"==", 20, false}, {"!=", 20, false}, {"<", 20, false}, {">", 20, false}, {">=", 20, false}, {"<=", 20, false},
            {"in", 20, false}, {"not in", 20, false}, {"contains", 20, false}, {"startsWith", 20, false}, {"endsWith", 20, false},
            {"+", 30, false}, {"-", 30, false}, {"*", 60, false}, {"/", 60, false}, {"%", 60, false},
            {"**", 100, true}, {"^", 100, true}, {"??", 500, false}};
        for (const auto& e : table)
            if (s == e.op) { *op = s; *info = {e.prec, e.right}; return true; }
        return false;
    }
    NodeP ex_expr(int prec, bool top) {
        Depth dd(depth_);
        if (too_deep()) return nullptr;
        NodeP l = ex_unary();
        if (!l) return nullptr;
        for (;;) {
            if (is_op("|") || is_op("..") || is_word("matches")) return refuse_op();
            std::string op;
            OpInfo info;
            if (!ex_binary(&op, &info) || info.prec < prec) break;
            adv();
            if (op == "not in") adv();
            NodeP r = ex_expr(info.right ? info.prec : info.prec + 1, false);
            if (!r) return nullptr;
            l = bin(op, std::move(l), std::move(r));
        }
        if (prec == 0 && is_op("?")) {
            adv();
            if (is_op(":")) return fail("the elvis operator '?:' is outside the supported subset");
            NodeP a = ex_expr(0, false);
            if (!a) return nullptr;
            if (!is_op(":")) return fail("expected ':' of the conditional before '" + show(cur()) + "'");
            adv();
            NodeP b = ex_expr(0, false);
            if (!b) return nullptr;
            NodeP n = mk(Node::Ternary);
            n->kids.push_back(std::move(l));
            n->kids.push_back(std::move(a));
            n->kids.push_back(std::move(b));
            return n;
        }
        (void)top;
        return l;
    }
    NodeP ex_unary() {
        if (is_op("-") || is_op("+") || is_op("!") || is_word("not")) {
            const std::string op = cur().s == "not" ? "!" : cur().s;
            const int prec = (op == "!") ? 50 : 90;
            adv();
            NodeP v = ex_expr(prec, false);
            if (!v) return nullptr;
            NodeP n = mk(Node::Unary, op);
            n->kids.push_back(std::move(v));
            return n;
        }
        return ex_postfix(ex_primary());
    }
    NodeP ex_primary() {
        const Tok t = cur();
        switch (t.k) {
            case Tok::Num: { adv(); NodeP n = mk(Node::Lit); n->lit = t.num; return n; }
            case Tok::Str: { adv(); NodeP n = mk(Node::Lit); n->lit = Str(t.s); return n; }
            case Tok::LParen: {
                adv();
                NodeP e = ex_expr(0, true);
                if (!e) return nullptr;
                if (cur().k != Tok::RParen) return fail("expected ')' before '" + show(cur()) + "'");
                adv();
                return e;
            }
            case Tok::LBrack: {
                adv();
                NodeP n = mk(Node::ListLit);
                if (cur().k == Tok::RBrack) { adv(); return n; }
                for (;;) {
                    NodeP e = ex_expr(0, true);
                    if (!e) return nullptr;
                    n->kids.push_back(std::move(e));
                    if (cur().k == Tok::Comma) { adv(); if (cur().k == Tok::RBrack) break; continue; }
                    break;
                }
                if (cur().k != Tok::RBrack) return fail("expected ']' before '" + show(cur()) + "'");
                adv();
                return n;
            }
            case Tok::Ident: {
                adv();
                if (t.s == "true" || t.s == "false") { NodeP n = mk(Node::Lit); n->lit = Bool(t.s == "true"); return n; }
                if (t.s == "nil") return mk(Node::Lit);
                if (t.s == "let") return fail("'let' is outside the supported subset");
                if (cur().k == Tok::LParen) return call(t.s);
                return mk(Node::Var, t.s);
            }
            case Tok::Op:
                if (t.s == "{") return fail("map literals are outside the supported subset");
                if (t.s == "#" || t.s == ".") return fail("pointer / lambda syntax is outside the supported subset");
                return fail("unexpected '" + show(t) + "'");
            default: return fail("unexpected '" + show(t) + "'");
        }
    }
    NodeP ex_postfix(NodeP base) {
        while (base) {
            if (is_op(".") || is_op("?.")) {
                const bool opt = cur().s == "?.";
                adv();
                if (cur().k != Tok::Ident) return fail("expected a member name after '.'");
                const std::string name = cur().s;
                adv();
                if (cur().k == Tok::LParen) return fail("method call '." + name + "(…)' is outside the supported subset");
                NodeP n = mk(Node::Member, name);
                n->optional = opt;
                n->kids.push_back(std::move(base));
                base = std::move(n);
            } else if (cur().k == Tok::LBrack) {
                adv();
                if (cur().k != Tok::Str) return fail("only a[\"name\"] indexing is inside the supported subset");
                const std::string name = cur().s;
                adv();
                if (cur().k != Tok::RBrack) return fail("only a[\"name\"] indexing is inside the supported subset");
                adv();
                NodeP n = mk(Node::Member, name);
                n->kids.push_back(std::move(base));
                base = std::move(n);
            } else {
                break;
            }
        }
        return base;
    }
};

// ---------------- evaluation
struct Eval {
    Dialect d;
    const V* params;
    std::string err;

    bool fail(const std::string& m) { if (err.empty()) err = m; return false; }

    // Go's == on two interface values of the kinds that occur here
    static bool go_equal(const V& a, const V& b, bool numeric_cross_type) {
        if (a.type != b.type) return false;
        switch (a.type) {
            case V::Null: return true;
            case V::Bool: return a.b == b.b;
            case V::String: return a.str == b.str;
            case V::Number:
                if (!numeric_cross_type && a.is_int != b.is_int) return false;
                if (a.is_int && b.is_int) return a.i == b.i;
                return a.num == b.num;
            case V::Array:
                if (a.arr.size() != b.arr.size()) return false;
                for (size_t i = 0; i < a.arr.size(); ++i) if (!go_equal(a.arr[i], b.arr[i], numeric_cross_type)) return false;
                return true;
            default: return false;
        }
    }
    // govaluate's parameter sanitiser: every integer kind becomes a float64
    static V sanitize(const V& v) { return (v.type == V::Number && v.is_int) ? Float(v.num) : v; }

    bool run(const Node* n, V* out) {
        switch (n->k) {
            case Node::Lit: *out = n->lit; return true;
            case Node::ListLit: {
                std::vector<V> xs;
                for (const auto& k : n->kids) { V v; if (!run(k.get(), &v)) return false; xs.push_back(std::move(v)); }
                *out = List(std::move(xs));
                return true;
            }
            case Node::Var: {
                const bool has = params && params->type == V::Object && params->obj.count(n->op);
                if (d == Dialect::Govaluate) {
                    if (!has) return fail("No parameter '" + n->op + "' found.");
                    *out = sanitize(params->obj.at(n->op));
                    return true;
                }
                *out = has ? params->obj.at(n->op) : Nil();        // expr.AllowUndefinedVariables
                return true;
            }
            case Node::Member: {
                V base;
                if (!run(n->kids[0].get(), &base)) return false;
                if (base.type == V::Null) {
                    if (n->optional) { *out = Nil(); return true; }
                    return fail("cannot fetch " + n->op + " from <nil>");
                }
                if (base.type != V::Object) return fail("cannot fetch " + n->op + " from a value that is not a map");
                auto it = base.obj.find(n->op);
                *out = it == base.obj.end() ? Nil() : it->second;
                return true;
            }
            case Node::Unary: {
                V v;
                if (!run(n->kids[0].get(), &v)) return false;
                if (n->op == "!") {
                    if (v.type != V::Bool) return fail("Value '" + sprint_v(v) + "' cannot be used with the logical prefix '!', it is not a bool");
                    *out = Bool(!v.b);
                    return true;
                }
                if (!is_num(v)) return fail("Value '" + sprint_v(v) + "' cannot be used with the numeric prefix '" + n->op + "', it is not a number");
                if (n->op == "+") { *out = v; return true; }
                *out = (d == Dialect::ExprLang && v.is_int) ? Int((long long)(0ull - (unsigned long long)v.i)) : Float(-v.num);
                return true;
            }
            case Node::Ternary: {
                V c;
                if (!run(n->kids[0].get(), &c)) return false;
                if (c.type != V::Bool) return fail("Value '" + sprint_v(c) + "' cannot be used with the ternary operator '?', it is not a bool");
                return run(n->kids[c.b ? 1 : 2].get(), out);
            }
            case Node::Call: {
                std::vector<V> args;
                for (const auto& k : n->kids) { V v; if (!run(k.get(), &v)) return false; args.push_back(std::move(v)); }
                if (d == Dialect::ExprLang && !HasFunction(n->op)) return builtin(n->op, args, out);
                std::string e;
                if (!CallFunction(n->op, args, out, &e)) return fail(e);
                return true;
            }
            case Node::Binary: return binary(n, out);
        }
        return fail("internal: unknown node");
    }

    bool builtin(const std::string& name, const std::vector<V>& a, V* out) {
        if (a.size() != 1) return fail("invalid number of arguments for " + name + " (expected 1, got " + std::to_string(a.size()) + ")");
        const V& x = a[0];
        if (name == "int") {
            if (is_num(x)) { *out = Int(x.is_int ? x.i : f2i(x.num)); return true; }
            if (is_str(x)) {
                char* e = nullptr;
                const long long r = strtoll(x.str.c_str(), &e, 10);
                if (x.str.empty() || *e != '\0') return fail("invalid operation: int(\"" + x.str + "\")");
                *out = Int(r);
                return true;
            }
            return fail("invalid operation: int(" + sprint_v(x) + ")");
        }
        if (name == "float") {
            if (is_num(x)) { *out = Float(x.num); return true; }
            if (is_str(x)) {
                char* e = nullptr;
                const double r = strtod(x.str.c_str(), &e);
                if (x.str.empty() || *e != '\0') return fail("invalid operation: float(\"" + x.str + "\")");
                *out = Float(r);
                return true;
            }
            return fail("invalid operation: float(" + sprint_v(x) + ")");
        }
        if (name == "string") { *out = Str(sprint_v(x)); return true; }
        if (name == "len") {
            if (is_str(x)) { *out = Int((long long)runes(x.str).size()); return true; }
            if (x.type == V::Array) { *out = Int((long long)x.arr.size()); return true; }
            if (x.type == V::Object) { *out = Int((long long)x.obj.size()); return true; }
            return fail("invalid argument for len (type " + std::string(x.type == V::Null ? "nil" : "scalar") + ")");
        }
        if (name == "abs") {
            if (!is_num(x)) return fail("invalid argument for abs");
            *out = x.is_int ? Int(x.i < 0 ? (long long)(0ull - (unsigned long long)x.i) : x.i) : Float(std::fabs(x.num));
            return true;
        }
        return fail("unknown function '" + name + "'");
    }

    bool binary(const Node* n, V* out) {
        const std::string& op = n->op;
        const bool gv = d == Dialect::Govaluate;
        V l, r;
        if (!run(n->kids[0].get(), &l)) return false;
        // short circuits (govaluate evaluationStage short-circuits && || ; expr-lang compiles jumps)
        if (op == "&&" || op == "and" || op == "||" || op == "or") {
            const bool is_and = op == "&&" || op == "and";
            if (l.type != V::Bool) return fail("Value '" + sprint_v(l) + "' cannot be used with the logical operator '" + op + "', it is not a bool");
            if (is_and ? !l.b : l.b) { *out = Bool(!is_and); return true; }
            if (!run(n->kids[1].get(), &r)) return false;
            if (r.type != V::Bool) return fail("Value '" + sprint_v(r) + "' cannot be used with the logical operator '" + op + "', it is not a bool");
            *out = Bool(r.b);
            return true;
        }
        if (op == "??") {
            if (l.type != V::Null) { *out = l; return true; }
            return run(n->kids[1].get(), out);
        }
        if (!run(n->kids[1].get(), &r)) return false;
        if (op == "==" || op == "!=") {                             // govaluate: reflect.DeepEqual; expr-lang: runtime.Equal
            const bool eq = go_equal(l, r, !gv);
            *out = Bool(op == "==" ? eq : !eq);
            return true;
        }
        if (op == "in" || op == "not in") {
            bool found = false;
            if (r.type == V::Array) {
                for (const V& e : r.arr) if (go_equal(l, gv ? sanitize(e) : e, !gv)) { found = true; break; }
            } else if (!gv && r.type == V::Object) {
                if (!is_str(l)) return fail("cannot use a non-string as a map key");
                found = r.obj.count(l.str) != 0;
            } else if (!gv && r.type == V::Null) {
                found = false;
            } else {
                return fail("Value '" + sprint_v(r) + "' cannot be used with the comparator 'in', it is not an array");
            }
            *out = Bool(op == "in" ? found : !found);
            return true;
        }
        if (op == "contains" || op == "startsWith" || op == "endsWith") {
            if (!is_str(l) || !is_str(r)) return fail("operator '" + op + "' needs two strings");
            bool b;
            if (op == "contains") b = l.str.find(r.str) != std::string::npos;
            else if (op == "startsWith") b = l.str.compare(0, r.str.size(), r.str) == 0;
            else b = l.str.size() >= r.str.size() && l.str.compare(l.str.size() - r.str.size(), r.str.size(), r.str) == 0;
            *out = Bool(b);
            return true;
        }
        if (op == "<" || op == "<=" || op == ">" || op == ">=") {
            int c;
            if (is_num(l) && is_num(r)) {
                if (l.is_int && r.is_int) c = l.i < r.i ? -1 : (l.i > r.i ? 1 : 0);
                else if (l.num != l.num || r.num != r.num) { *out = Bool(false); return true; }
                else c = l.num < r.num ? -1 : (l.num > r.num ? 1 : 0);
            } else if (is_str(l) && is_str(r)) {
                c = l.str.compare(r.str);
                c = c < 0 ? -1 : (c > 0 ? 1 : 0);
            } else {
                return fail("Value '" + sprint_v(is_num(l) || is_str(l) ? r : l) + "' cannot be used with the comparator '" + op + "', it is not a number");
            }
            *out = Bool(op == "<" ? c < 0 : op == "<=" ? c <= 0 : op == ">" ? c > 0 : c >= 0);
            return true;
        }
        if (op == "+") {
            if (gv) {
                if (is_str(l) || is_str(r)) { *out = Str(sprint_v(l) + sprint_v(r)); return true; }   // fmt.Sprintf("%v%v")
            } else if (is_str(l) && is_str(r)) {
                *out = Str(l.str + r.str);
                return true;
            }
        }
        if (!is_num(l) || !is_num(r))
            return fail("Value '" + sprint_v(is_num(l) ? r : l) + "' cannot be used with the modifier '" + op + "', it is not a number");
        const bool ints = !gv && l.is_int && r.is_int;
        if (op == "+") { *out = ints ? Int((long long)((unsigned long long)l.i + (unsigned long long)r.i)) : Float(l.num + r.num); return true; }
        if (op == "-") { *out = ints ? Int((long long)((unsigned long long)l.i - (unsigned long long)r.i)) : Float(l.num - r.num); return true; }
        if (op == "*") { *out = ints ? Int((long long)((unsigned long long)l.i * (unsigned long long)r.i)) : Float(l.num * r.num); return true; }
        if (op == "/") { *out = Float(l.num / r.num); return true; }
        if (op == "%") {
            if (gv) { *out = Float(std::fmod(l.num, r.num)); return true; }          // math.Mod
            if (!ints) return fail("invalid operation: operator % needs two integers");
            if (r.i == 0) return fail("integer divide by zero");
            *out = Int(r.i == -1 ? 0 : l.i % r.i);
            return true;
        }
        if (op == "**" || op == "^") { *out = Float(std::pow(l.num, r.num)); return true; }
        return fail("operator '" + op + "' is outside the supported subset");
    }
};

struct Program {
    Dialect d;
    NodeP root;
    bool Run(const V& params, V* out, std::string* err) const {
        Eval e{d, &params, ""};
        if (!e.run(root.get(), out)) { *err = e.err; return false; }
        return true;
    }
};

static std::shared_ptr<Program> compile(const std::string& src, Dialect d, std::string* err) {
    std::vector<Tok> toks;
    std::string e;
    if (!Lexer(src, d).Run(&toks, &e)) { *err = e; return nullptr; }
    Parser p(std::move(toks), d);
    NodeP root = p.Parse(&e);
    if (!root) { *err = e; return nullptr; }
    auto prog = std::make_shared<Program>();
    prog->d = d;
    prog->root = std::move(root);
    return prog;
}

// ---------------- the normalizers (normalizer.go:43-171)
struct HourNormalizer : Normalizer {
    V Apply(const V&) override { return Int(local_now().tm_hour); }
    const char* Kind() const override { return "CreateHourNormalizer"; }
};
struct DayNormalizer : Normalizer {                                 // Monday 0 … Sunday 6 (:51-68)
    V Apply(const V&) override { const int w = local_now().tm_wday; return Int(w == 0 ? 6 : w - 1); }
    const char* Kind() const override { return "CreateDayNormalizer"; }
};
struct MonthNormalizer : Normalizer {
    V Apply(const V&) override { return Int(local_now().tm_mon + 1); }
    const char* Kind() const override { return "CreateMonthNormalizer"; }
};
struct WeekNormalizer : Normalizer {
    V Apply(const V&) override { return Int(iso_week(local_now())); }
    const char* Kind() const override { return "CreateWeekNormalizer"; }
};
struct RandomNormalizer : Normalizer {                              // rand.Intn(100) (:86-97): any value of [0, 100)
    std::atomic<uint64_t> state;                                    // (requests run concurrently; math/rand's source is locked)
    RandomNormalizer() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); state = (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
    V Apply(const V&) override {
        uint64_t z = state.fetch_add(0x9E3779B97F4A7C15ull) + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return Int((long long)(z % 100));
    }
    const char* Kind() const override { return "CreateRandomNormalizer"; }
};
struct ConstValueNormalizer : Normalizer {                          // Apply returns nil; the op stores the feature's source (:99-110)
    V Apply(const V&) override { return Nil(); }
    const char* Kind() const override { return "CreateConstValueNormalizer"; }
};
struct ProgramNormalizer : Normalizer {                             // ExpressionNormalizer / ExprNormalizer
    std::shared_ptr<Program> prog;
    bool is_expr;
    V Apply(const V& value) override {                              // :126-138 / :157-171: anything but a result is ""
        if (!prog || value.type != V::Object) return Str("");
        V out;
        std::string e;
        if (!prog->Run(value, &out, &e)) return Str("");
        return out;
    }
    const char* Kind() const override { return is_expr ? "ExprNormalizer" : "ExpressionNormalizer"; }
};
}  // namespace

std::shared_ptr<Normalizer> NewNormalizer(const std::string& name, const std::string& expression, std::string* err) {
    if (name == "hour_in_day") return std::make_shared<HourNormalizer>();
    if (name == "weekday") return std::make_shared<DayNormalizer>();
    if (name == "random") return std::make_shared<RandomNormalizer>();
    if (name == "const_value") return std::make_shared<ConstValueNormalizer>();
    if (name == "month") return std::make_shared<MonthNormalizer>();
    if (name == "week") return std::make_shared<WeekNormalizer>();
    if (name == "expression" || name == "expr") {
        std::string e;
        auto prog = compile(expression, name == "expr" ? Dialect::ExprLang : Dialect::Govaluate, &e);
        if (!prog) {
            if (err) *err = "normalizer \"" + name + "\": `" + expression + "`: " + e;
            return nullptr;
        }
        auto n = std::make_shared<ProgramNormalizer>();
        n->prog = prog;
        n->is_expr = name == "expr";
        return n;
    }
    return nullptr;                                                 // normalizer.go:21-40: an unknown name leaves the interface nil
}

// ---- properties ----------------------------------------------------------------------------------------------
std::string ItemStringProperty(const module::Item& it, const std::string& key) {          // item.go:101-122
    auto p = it.Properties.find(key);
    if (p == it.Properties.end()) return "";
    const V& v = p->second;
    if (v.type == V::String) return v.str;
    if (v.type == V::Number && !v.is_u64) return std::to_string(v.is_int ? v.i : f2i(v.num));   // float64: strconv.Itoa(int(value))
    return "";
}
std::string UserStringProperty(const module::User& u, const std::string& key) {           // user.go:168-189
    auto p = u.Properties.find(key);
    if (p == u.Properties.end()) return "";
    const V& v = p->second;
    if (v.type == V::String) return v.str;
    if (v.type == V::Number && !v.is_u64) return v.is_int ? std::to_string(v.i) : go_format_f(v.num);
    return "";
}

static std::vector<std::string> split(const std::string& s, char sep) {                   // strings.Split
    std::vector<std::string> out;
    size_t pos = 0;
    for (;;) {
        const size_t hit = s.find(sep, pos);
        if (hit == std::string::npos) { out.push_back(s.substr(pos)); return out; }
        out.push_back(s.substr(pos, hit - pos));
        pos = hit + 1;
    }
}
static V object_of(const std::map<std::string, V>& props) {
    V o;
    o.type = V::Object;
    o.obj = props;
    return o;
}
static V bool_as_01(const V& r) { return r.type == V::Bool ? Int(r.b ? 1 : 0) : r; }      // new_feature_op.go:23-31,104-112

// ---- Feature -------------------------------------------------------------------------------------------------
bool Feature::LoadWithConfig(const std::vector<FeatureConfig>& features, std::string* err) {
    trans_.clear();
    for (const FeatureConfig& c : features) {
        static const char* const types[] = {"raw_feature", "compose_feature", "delete_feature", "batch_raw_feature", "new_feature", "context_feature"};
        bool known = false;
        for (const char* t : types) known = known || c.FeatureType == t;
        if (!known) { if (err) *err = "not find feature type:" + c.FeatureType; return false; }      // op.go:32 panics
        Trans t;
        t.conf = c;
        t.source = c.FeatureValue.empty() ? c.FeatureSource : c.FeatureValue;                        // feature.go:24-27
        std::string e;
        t.normalizer = NewNormalizer(c.Normalizer, c.Expression, &e);
        if (!t.normalizer && !e.empty()) { if (err) *err = "feature \"" + c.FeatureName + "\": " + e; return false; }
        trans_.push_back(std::move(t));
    }
    return true;
}

void Feature::LoadFeatures(module::User* user, std::vector<module::ItemPtr>& items, context::RecommendContext* ctx) {
    static const module::User no_user;
    V user_params_cache;                                            // Context_User_Features_Key (new_feature_op.go:50-52,66-76)
    bool have_user_params = false;
    for (const Trans& t : trans_) {
        const std::string& type = t.conf.FeatureType;
        const std::string& name = t.conf.FeatureName;
        const std::string& source = t.source;
        const bool remove = t.conf.RemoveFeatureSource;
        Normalizer* nz = t.normalizer.get();
        const std::string kind = nz ? nz->Kind() : "";
        if (t.conf.FeatureStore == "item") {                        // feature.go:73-78
            for (auto& item : items) {
                if (type == "raw_feature") {                        // op.go:69-92
                    const auto c = split(source, ':');
                    if (c.size() < 2) continue;
                    if (c[0] == "user") {
                        item->AddProperty(name, Str(UserStringProperty(user ? *user : no_user, c[1])));
                        if (remove && user) user->Properties.erase(c[1]);
                    } else {
                        const std::string value = ItemStringProperty(*item, c[1]);
                        item->AddProperty(name, nz ? nz->Apply(Str(value)) : Str(value));
                        if (remove) item->Properties.erase(c[1]);
                    }
                } else if (type == "compose_feature") {             // op.go:116-143
                    std::string value = name;
                    for (const std::string& val : split(source, ',')) {
                        const auto c = split(val, ':');
                        if (c.size() < 2) continue;
                        if (c[0] == "user") {
                            value += "_" + UserStringProperty(user ? *user : no_user, c[1]);
                        } else {
                            value += "_" + (c[1] == "id" ? item->Id : ItemStringProperty(*item, c[1]));
                            if (remove) item->Properties.erase(c[1]);
                        }
                    }
                    item->AddProperty(name, Str(value));
                } else if (type == "delete_feature") {              // delete_feature_op.go:32-44
                    for (const std::string& val : split(source, ',')) {
                        const auto c = split(val, ':');
                        item->Properties.erase(c.size() >= 2 ? c[1] : c[0]);
                    }
                } else if (type == "batch_raw_feature") {           // batch_raw_feature_op.go:37-58
                    const auto names = split(name, ','), sources = split(source, ',');
                    if (names.size() != sources.size()) continue;
                    for (size_t i = 0; i < sources.size(); ++i) {
                        const auto c = split(sources[i], ':');
                        if (c.size() < 2) continue;
                        item->AddProperty(names[i], Str(c[0] == "user" ? UserStringProperty(user ? *user : no_user, c[1])
                                                                        : ItemStringProperty(*item, c[1])));
                    }
                } else if (type == "new_feature") {                 // new_feature_op.go:54-115
                    if (!nz) continue;                              // (a nil normalizer is a nil-pointer panic there)
                    const long long now_s = now_ms() / 1000;
                    V params;
                    params.type = V::Object;
                    params.obj["currentTime"] = Int(now_s);
                    if (source == "item:recall_name") {
                        params.obj["recall_name"] = Str(item->RetrieveId);
                    } else if (source.empty()) {
                        V item_params = object_of(item->Properties);
                        if (!item_params.obj.count("recall_name")) item_params.obj["recall_name"] = Str(item->RetrieveId);
                        if (kind == "ExprNormalizer") {
                            if (user && !have_user_params) { user_params_cache = object_of(user->Properties); have_user_params = true; }
                            params.obj.clear();
                            params.obj["item"] = std::move(item_params);
                            params.obj["user"] = user ? user_params_cache : object_of(std::map<std::string, V>());
                            params.obj["currentTime"] = Int(now_s);
                        } else {
                            params = std::move(item_params);
                            params.obj["currentTime"] = Int(now_s);
                        }
                    } else {
                        const auto c = split(source, ':');
                        if (c.size() >= 2) {
                            const auto& props = c[0] == "user" ? (user ? user->Properties : no_user.Properties) : item->Properties;
                            auto p = props.find(c[1]);
                            params.obj[c[1]] = p == props.end() ? Nil() : p->second;
                        }
                    }
                    item->AddProperty(name, bool_as_01(nz->Apply(params)));
                }
                // context_feature: ItemTransOp is empty (op.go:160-161)
            }
            continue;
        }
        if (!user) continue;                                        // the reference dereferences user; a nil user is a panic there
        if (type == "raw_feature") {                                // op.go:55-64
            const auto c = split(source, ':');
            if (c.size() >= 2) {
                user->Properties[name] = Str(UserStringProperty(*user, c[1]));
                if (remove) user->Properties.erase(c[1]);
            }
        } else if (type == "compose_feature") {                     // op.go:100-114
            std::string value;
            for (const std::string& val : split(source, ',')) {
                const auto c = split(val, ':');
                if (c.size() < 2) continue;
                value += "_" + UserStringProperty(*user, c[1]);
                if (remove) user->Properties.erase(c[1]);
            }
            user->Properties[name] = Str(value);
        } else if (type == "delete_feature") {                      // delete_feature_op.go:16-28
            for (const std::string& val : split(source, ',')) {
                const auto c = split(val, ':');
                user->Properties.erase(c.size() >= 2 ? c[1] : c[0]);
            }
        } else if (type == "batch_raw_feature") {                   // batch_raw_feature_op.go:18-33
            const auto names = split(name, ','), sources = split(source, ',');
            if (names.size() != sources.size()) continue;
            for (size_t i = 0; i < sources.size(); ++i) {
                const auto c = split(sources[i], ':');
                if (c.size() >= 2) user->Properties[names[i]] = Str(UserStringProperty(*user, c[1]));
            }
        } else if (type == "new_feature") {                         // new_feature_op.go:16-48
            if (!nz) continue;
            if (kind == "CreateConstValueNormalizer") {
                user->Properties[name] = Str(source);
            } else if (kind == "ExpressionNormalizer") {
                user->Properties[name] = bool_as_01(nz->Apply(object_of(user->Properties)));
            } else if (kind == "ExprNormalizer") {
                V params;
                params.type = V::Object;
                params.obj["user"] = object_of(user->Properties);
                params.obj["currentTime"] = Int(now_ms() / 1000);
                user->Properties[name] = bool_as_01(nz->Apply(params));
            } else {
                user->Properties[name] = nz->Apply(Nil());
            }
        } else if (type == "context_feature") {                     // op.go:150-158: the request's "features" object
            if (ctx) {
                auto f = ctx->Param.find("features");
                if (f != ctx->Param.end() && f->second.type == V::Object)
                    for (const auto& kv : f->second.obj) user->Properties[kv.first] = kv.second;
            }
        }
    }
}

}  // namespace feature
}  // namespace pairec
