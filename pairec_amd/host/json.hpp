// json.hpp — minimal JSON value + parser/serializer for the recconf subset and the test driver.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace pairec {
namespace json {

struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    bool is_int = false;
    bool is_u64 = false;           // an integer that is a Go uint64 (utils "hash"): `i` holds its bit pattern
    long long i = 0;
    std::string str;
    std::vector<Value> arr;
    std::map<std::string, Value> obj;

    Value() = default;
    static Value Num(double d) { Value v; v.type = Number; v.num = d; return v; }
    static Value Str(std::string s) { Value v; v.type = String; v.str = std::move(s); return v; }
    bool has(const std::string& k) const { return type == Object && obj.count(k) != 0; }
    const Value& at(const std::string& k) const { static const Value nul; auto it = obj.find(k); return it == obj.end() ? nul : it->second; }
    std::string s(const std::string& k, const std::string& def = "") const { const Value& v = at(k); return v.type == String ? v.str : def; }
    double d(const std::string& k, double def = 0.0) const { const Value& v = at(k); return v.type == Number ? v.num : def; }
    long long n(const std::string& k, long long def = 0) const { const Value& v = at(k); return v.type == Number ? (long long)v.num : def; }
};

class Parser {
public:
    explicit Parser(const std::string& t) : s_(t) {}
    bool Parse(Value* out, std::string* err) {
        ws();
        if (!value(out)) { if (err) *err = "json: parse error at offset " + std::to_string(p_); return false; }
        ws();
        if (p_ != s_.size()) { if (err) *err = "json: trailing characters at offset " + std::to_string(p_); return false; }
        return true;
    }
private:
    const std::string& s_;
    size_t p_ = 0;
    int depth_ = 0;
    // nesting bound (encoding/json stops at 10 000 levels: "exceeded max depth"; this parser recurses on the C++ stack — a
    // config of a million '[' must be an error, not a stack overflow)
    static constexpr int kMaxDepth = 1000;
    struct Nest {
        int& d;
        explicit Nest(int& depth) : d(depth) { ++d; }
        ~Nest() { --d; }
    };
    void ws() { while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) ++p_; }
    bool lit(const char* w) { size_t n = strlen(w); if (s_.compare(p_, n, w) == 0) { p_ += n; return true; } return false; }
    bool value(Value* v) {
        if (p_ >= s_.size()) return false;
        const char c = s_[p_];
        if (c == '{') return object(v);
        if (c == '[') return array(v);
        if (c == '"') { v->type = Value::String; return string(&v->str); }
        if (lit("true")) { v->type = Value::Bool; v->b = true; return true; }
        if (lit("false")) { v->type = Value::Bool; v->b = false; return true; }
        if (lit("null")) { v->type = Value::Null; return true; }
        return number(v);
    }
    bool number(Value* v) {
        const char* b = s_.c_str() + p_;
        char* e = nullptr;
        const double d = strtod(b, &e);
        if (e == b) return false;
        v->type = Value::Number;
        v->num = d;
        bool isint = true;
        for (const char* q = b; q < e; ++q) if (*q == '.' || *q == 'e' || *q == 'E') isint = false;
        v->is_int = isint;
        if (isint) v->i = strtoll(b, nullptr, 10);
        p_ += (size_t)(e - b);
        return true;
    }
    bool string(std::string* out) {
        ++p_;
        out->clear();
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c == '\\' && p_ < s_.size()) {
                const char e = s_[p_++];
                switch (e) {
                    case 'n': c = '\n'; break;
                    case 't': c = '\t'; break;
                    case 'r': c = '\r'; break;
                    case 'b': c = '\b'; break;
                    case 'f': c = '\f'; break;
                    case 'u': {
                        if (p_ + 4 > s_.size()) return false;
                        const unsigned cp = (unsigned)strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                        p_ += 4;
                        if (cp < 0x80) out->push_back((char)cp);
                        else if (cp < 0x800) { out->push_back((char)(0xC0 | (cp >> 6))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                        else { out->push_back((char)(0xE0 | (cp >> 12))); out->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out->push_back((char)(0x80 | (cp & 0x3F))); }
                        continue;
                    }
                    default: c = e;
                }
            }
            out->push_back(c);
        }
        if (p_ >= s_.size()) return false;
        ++p_;
        return true;
    }
    bool array(Value* v) {
        Nest nest(depth_);
        if (depth_ > kMaxDepth) return false;
        v->type = Value::Array;
        ++p_;
        ws();
        if (p_ < s_.size() && s_[p_] == ']') { ++p_; return true; }
        for (;;) {
            Value e;
            ws();
            if (!value(&e)) return false;
            v->arr.push_back(std::move(e));
            ws();
            if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
            if (p_ < s_.size() && s_[p_] == ']') { ++p_; return true; }
            return false;
        }
    }
    bool object(Value* v) {
        Nest nest(depth_);
        if (depth_ > kMaxDepth) return false;
        v->type = Value::Object;
        ++p_;
        ws();
        if (p_ < s_.size() && s_[p_] == '}') { ++p_; return true; }
        for (;;) {
            ws();
            if (p_ >= s_.size() || s_[p_] != '"') return false;
            std::string k;
            if (!string(&k)) return false;
            ws();
            if (p_ >= s_.size() || s_[p_] != ':') return false;
            ++p_;
            ws();
            Value e;
            if (!value(&e)) return false;
            v->obj[k] = std::move(e);
            ws();
            if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
            if (p_ < s_.size() && s_[p_] == '}') { ++p_; return true; }
            return false;
        }
    }
};

inline void Escape(const std::string& s, std::string* out) {
    out->push_back('"');
    for (char c : s) {
        switch (c) {
            case '"': *out += "\\\""; break;
            case '\\': *out += "\\\\"; break;
            case '\n': *out += "\\n"; break;
            case '\t': *out += "\\t"; break;
            case '\r': *out += "\\r"; break;
            default: out->push_back(c);
        }
    }
    out->push_back('"');
}

inline std::string NumToString(double d) {
    char buf[40];
    snprintf(buf, sizeof buf, "%.17g", d);
    return buf;
}

}  // namespace json
}  // namespace pairec
