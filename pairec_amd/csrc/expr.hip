// expr.hip — RankConfig.RankScore score fusion: compile on the host, evaluate per item on device.
//
// Replaces ast.GetExpAST / ExprASTResult (utils/ast/ast.go:215-268,368-389; lexer
// utils/ast/parse.go:40-158), evaluated once per candidate in RankService.Rank
// (service/rank/rank_service.go:339-363).  Grammar and quirks follow the reference:
//   operators  # ( ) + - * / ^ %   with precedence {+,-:20  *,/,%:40  ^:60  #:80} (ast.go:81),
//   all binary operators left-associative (parseBinOpRHS, ast.go:169-197);
//   literals start with a digit and extend over [0-9._e] ('_' removed); `${name}` parameters;
//   a malformed literal or an operator in operand position yields the number 0 WITHOUT consuming
//   the token (parseNumber, ast.go:108-124) and the recorded error is ignored by GetExpAST — so
//   "-5" is 0-5 and "2*1e-5" is 2*0;  `#` = first non-zero, `^` = math.Pow, `%` = int(l) % int(r);
//   `/` by zero and `%` by zero panic in the reference → PG_ERR_ARITH here.
// ASTType "antlr" (GetExpASTByAntlr / ExprASTResultByAntlr, ast.go:275-389) hands the source to go-antlr-valuate v0.0.4, which
// is not vendored.  pg_expr_compile_typed(…, "antlr") serves the SUBSET of that language the reference's own tests pin
// (ast_test.go:30-56,90-167,213-300): + - * / ^ with ^ = math.Pow above * / above + -, parentheses, unary minus (not against ^), numbers,
// ${name}, and the registered functions maxIndex(${v}) / maxValue(${v}) over a list property (antlr_functions.go:34-66) —
// compiled to the same device program; anything else is refused BY NAME (PG_ERR_UNSUPPORTED), never evaluated differently.
#include "common.hpp"

#include <cmath>
#include <string>
#include <vector>

namespace pg {

enum OpCode : uint32_t { OP_CONST = 0, OP_VAR, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD, OP_POW, OP_FNZ,
                         OP_DIVF,   // antlr subset: float division as Go's `/` on float64 (no panic: ±Inf / NaN)
                         OP_NEG };  // antlr subset: unary minus

struct Instr {
    uint32_t op;
    uint32_t arg;     // variable index
    double val;       // constant
};

constexpr int kMaxStack = 32;
constexpr int kMaxProg = 128;

}  // namespace pg

struct pg_expr {
    std::string source;
    std::vector<pg::Instr> prog;
    std::vector<std::string> vars;
    int max_depth = 0;
    bool empty = false;       // "" → no expression (GetExpAST returns nil)
    bool antlr = false;       // compiled by pg_expr_compile_typed(…, "antlr"): the evaluation-error rule of ExprASTResultByAntlr applies on the host
    // RankConfig.ScoreRewrite of the scene this RankScore belongs to (pg_expr_set_score_rewrites): evaluated by the
    // recommend pipelines' fusion stage before the RankScore itself (pipeline.hip: post_fuse_sort_locked)
    struct Rewrite {
        std::string source;
        bool failed = false;  // the source's expression did not compile in the reference: the score is 0 (rank_service.go:349-351)
        std::vector<pg::Instr> prog;
        std::vector<std::string> vars;
    };
    std::vector<Rewrite> rewrites;
    mutable std::atomic<int> holders{0};     // bindings made from this expression that are still alive (pg::ExprHold)
};

namespace pg {

// ---- lexer (parse.go:40-158) -----------------------------------------------------------------
enum TokType { T_LITERAL = 0, T_OPERATOR = 1, T_PARAMETER = 2 };
struct Token {
    std::string tok;
    TokType type;
};

static bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }
static bool is_lit(char c) { return (c >= '0' && c <= '9') || c == '.' || c == '_' || c == 'e'; }

static int tokenize(const std::string& s, std::vector<Token>* out) {
    size_t off = 0;
    const size_t n = s.size();
    while (off < n) {
        char last_ws = 0;
        while (off < n && is_ws(s[off])) last_ws = s[off++];
        if (off >= n) {
            // the reference re-examines its stale `ch`: only a plain space is tolerated
            if (last_ws != 0 && last_ws != ' ') {
                set_error("pg_expr_compile: symbol error: unknown trailing whitespace 0x%02x", last_ws);
                return PG_ERR_PARSE;
            }
            break;
        }
        const char ch = s[off];
        const size_t start = off;
        if (ch == '#' || ch == '(' || ch == ')' || ch == '+' || ch == '-' || ch == '*' || ch == '/' ||
            ch == '^' || ch == '%') {
            out->push_back({std::string(1, ch), T_OPERATOR});
            ++off;
        } else if (ch >= '0' && ch <= '9') {
            while (off < n && is_lit(s[off])) ++off;
            std::string t;
            for (size_t i = start; i < off; ++i)
                if (s[i] != '_') t.push_back(s[i]);
            out->push_back({t, T_LITERAL});
        } else if (ch == '$') {
            ++off;
            if (off < n && s[off] == '{') {
                while (off < n && s[off] != '}') ++off;
                out->push_back({s.substr(start + 2, off - (start + 2)), T_PARAMETER});
                ++off;
            } else {
                break;       // nil token: lexing stops silently (parse.go:112-124)
            }
        } else {
            set_error("pg_expr_compile: symbol error: unknown '%c', pos [%zu:]", ch, start);
            return PG_ERR_PARSE;
        }
    }
    return PG_OK;
}

// strconv.ParseFloat over the literal alphabet; false where Go reports an error
static bool go_parse_float(const std::string& t, double* v) {
    size_t i = 0;
    const size_t n = t.size();
    size_t nd = 0;
    while (i < n && t[i] >= '0' && t[i] <= '9') { ++i; ++nd; }
    if (nd == 0) return false;
    if (i < n && t[i] == '.') {
        ++i;
        while (i < n && t[i] >= '0' && t[i] <= '9') ++i;
    }
    if (i < n && t[i] == 'e') {
        ++i;
        size_t ne = 0;
        while (i < n && t[i] >= '0' && t[i] <= '9') { ++i; ++ne; }
        if (ne == 0) return false;
    }
    if (i != n) return false;
    const double d = strtod(t.c_str(), nullptr);
    if (std::isinf(d)) return false;        // ErrRange
    *v = d;
    return true;
}

// ---- parser (ast.go:84-197) → postfix program ------------------------------------------------
struct Node {
    int kind;            // 0 number, 1 parameter, 2 binary
    double val = 0;
    std::string name;
    char op = 0;
    int lhs = -1, rhs = -1;
};

struct Parser {
    const std::vector<Token>& toks;
    size_t i = 0;
    std::vector<Node> nodes;
    explicit Parser(const std::vector<Token>& t) : toks(t) {}
    const Token& cur() const { return toks[i < toks.size() ? i : toks.size() - 1]; }   // stale currTok
    bool next() {
        i = i + 1 < toks.size() + 1 ? i + 1 : toks.size();
        return i < toks.size();
    }
    int prec() const {
        const std::string& t = cur().tok;          // keyed on the text only (ast.go:100-105)
        if (t == "+" || t == "-") return 20;
        if (t == "*" || t == "/" || t == "%") return 40;
        if (t == "^") return 60;
        if (t == "#") return 80;
        return -1;
    }
    int add(Node n) {
        nodes.push_back(std::move(n));
        return (int)nodes.size() - 1;
    }
    int parse_number() {
        double v;
        if (!go_parse_float(cur().tok, &v)) {
            Node n;
            n.kind = 0;
            n.val = 0.0;
            return add(n);                          // error recorded and ignored; token NOT consumed
        }
        next();
        Node n;
        n.kind = 0;
        n.val = v;
        return add(n);
    }
    int parse_primary() {
        const Token t = cur();
        if (t.type == T_LITERAL) return parse_number();
        if (t.type == T_PARAMETER) {
            next();
            Node n;
            n.kind = 1;
            n.name = t.tok;
            return add(n);
        }
        if (t.tok == "(") {
            next();
            const int e = parse_expression();
            if (e < 0) return -1;
            if (cur().tok != ")") return -1;        // "want ')'" → nil
            next();
            return e;
        }
        return parse_number();
    }
    int parse_expression() {
        const int lhs = parse_primary();
        return parse_binop_rhs(0, lhs);
    }
    int parse_binop_rhs(int exec_prec, int lhs) {
        // NOTE the reference calls parseBinOpRHS(0, nil) when parsePrimary fails; lhs<0 propagates
        for (;;) {
            const int tp = prec();
            if (tp < exec_prec) return lhs;
            const char op = cur().tok[0];
            if (!next()) return lhs;
            int rhs = parse_primary();
            if (rhs < 0) return -1;
            if (tp < prec()) {
                rhs = parse_binop_rhs(tp + 1, rhs);
                if (rhs < 0) return -1;
            }
            Node n;
            n.kind = 2;
            n.op = op;
            n.lhs = lhs;
            n.rhs = rhs;
            lhs = add(n);
        }
    }
};

static int emit(const Parser& p, int node, pg_expr* e, int depth, int* max_depth) {
    if (node < 0) {
        // nil sub-tree: ExprASTResult falls through to `return 0.0`
        e->prog.push_back({OP_CONST, 0, 0.0});
        *max_depth = std::max(*max_depth, depth + 1);
        return PG_OK;
    }
    const Node& n = p.nodes[node];
    if (n.kind == 0) {
        e->prog.push_back({OP_CONST, 0, n.val});
        *max_depth = std::max(*max_depth, depth + 1);
    } else if (n.kind == 1) {
        uint32_t idx = 0;
        for (; idx < e->vars.size(); ++idx)
            if (e->vars[idx] == n.name) break;
        if (idx == e->vars.size()) e->vars.push_back(n.name);
        e->prog.push_back({OP_VAR, idx, 0.0});
        *max_depth = std::max(*max_depth, depth + 1);
    } else {
        int rc;
        const size_t mark = e->prog.size();
        if ((rc = emit(p, n.lhs, e, depth, max_depth))) return rc;
        if ((rc = emit(p, n.rhs, e, depth + 1, max_depth))) return rc;
        uint32_t op;
        switch (n.op) {
            case '+': op = OP_ADD; break;
            case '-': op = OP_SUB; break;
            case '*': op = OP_MUL; break;
            case '/': op = OP_DIV; break;
            case '%': op = OP_MOD; break;
            case '^': op = OP_POW; break;
            case '#': op = OP_FNZ; break;
            default:
                // unknown operator: ExprASTResult's default branch → 0.0 (operands have no side
                // effects, so the whole sub-tree collapses to the constant)
                e->prog.resize(mark);
                e->prog.push_back({OP_CONST, 0, 0.0});
                return PG_OK;
        }
        e->prog.push_back({op, 0, 0.0});
    }
    return PG_OK;
}

struct ExprDev {
    Instr prog[kMaxProg];
    uint32_t n;
};

// math.Pow as `^` sees it (utils/ast/ast.go:246; Go stdlib math/pow.go, go 1.24 per the reference's go.mod).  Go does not call a
// libm pow: Pow(x, 1) = x and Pow(x, +-0.5) = Sqrt(x), 1 / Sqrt(x) are exact special cases, and the INTEGER part of the exponent
// is applied by repeated squaring of Frexp(x)'s mantissa with the binary exponent carried on the side — so 400^4 is exactly
// 25 600 000 000 where pow() is an ulp off (and that ulp decides whether the power is an integer-valued exponent of the next
// `^`, or what an integer `%` of it leaves: found by scripts/soak_expr.py).  Integer-valued exponents therefore take Go's loop
// here, bit for bit (the oracle restates the same loop); fractional ones stay on pow(), within 2 ulp of Go's Exp(yf Log(x)) form.
__device__ __forceinline__ double go_pow(double x, double y) {
    if (y == 1.0) return x;
    const bool xfin = x == x && fabs(x) != __builtin_inf();
    if (y == 0.5 && xfin && x != 0.0) return sqrt(x);
    if (y == -0.5 && xfin && x != 0.0) return 1.0 / sqrt(x);
    const double ay = fabs(y);
    if (xfin && x != 0.0 && x != 1.0 && y != 0.0 && ay < 9223372036854775808.0 && ay == trunc(ay)) {
        double a1 = 1.0;
        long long ae = 0;
        int xe_i;
        double x1 = frexp(x, &xe_i);
        long long xe = xe_i;
        for (long long i = (long long)ay; i != 0; i >>= 1) {
            if (xe < -(1ll << 12) || (1ll << 12) < xe) {
                // overflow / underflow of the result: catch the exponent, stop
                ae += xe;
                break;
            }
            if (i & 1) {
                a1 *= x1;
                ae += xe;
            }
            x1 *= x1;
            xe <<= 1;
            if (x1 < 0.5) {
                x1 += x1;
                xe--;
            }
        }
        if (y < 0.0) {
            a1 = 1.0 / a1;
            ae = -ae;
        }
        if (ae > 4096) ae = 4096;                     // ldexp's int argument: far beyond the format either way
        if (ae < -4096) ae = -4096;
        return ldexp(a1, (int)ae);
    }
    return pow(x, y);
}

__global__ void expr_eval_kernel(ExprDev e, const double* __restrict__ vars, uint32_t n_items,
                                 double* __restrict__ out, uint32_t* __restrict__ err, uint32_t items_per_flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    double st[kMaxStack];
    int sp = 0;
    bool bad = false;
    for (uint32_t pc = 0; pc < e.n; ++pc) {
        const Instr in = e.prog[pc];
        if (in.op == OP_CONST) {
            st[sp++] = in.val;
        } else if (in.op == OP_VAR) {
            st[sp++] = vars[(size_t)in.arg * n_items + i];
        } else {
            if (in.op == OP_NEG) {
                st[sp - 1] = -st[sp - 1];
                continue;
            }
            const double r = st[--sp];
            const double l = st[--sp];
            double v = 0.0;
            switch (in.op) {
                case OP_DIVF: v = l / r; break;
                case OP_ADD: v = l + r; break;
                case OP_SUB: v = l - r; break;
                case OP_MUL: v = l * r; break;
                case OP_DIV:
                    if (r == 0.0) bad = true; else v = l / r;
                    break;
                case OP_MOD: {
                    // float64(int(l) % int(r)); Go's float→int of NaN/out-of-range gives MinInt64 on amd64
                    const long long li = (l == l && fabs(l) < 9223372036854775808.0) ? (long long)l : (long long)0x8000000000000000ull;
                    const long long ri = (r == r && fabs(r) < 9223372036854775808.0) ? (long long)r : (long long)0x8000000000000000ull;
                    if (ri == 0) bad = true;
                    else if (ri == -1) v = 0.0;
                    else v = (double)(li % ri);
                    break;
                }
                case OP_POW: v = go_pow(l, r); break;
                case OP_FNZ: v = (l != 0.0) ? l : r; break;
            }
            st[sp++] = v;
        }
    }
    out[i] = sp > 0 ? st[sp - 1] : 0.0;
    if (bad) atomicOr(err + (items_per_flag ? i / items_per_flag : 0u), 1u);
}

int expr_eval_enqueue_locked(pg_ctx* ctx, const pg_expr* e, const double* d_vars, uint32_t n_items, double* d_out,
                             uint32_t* d_err, uint32_t items_per_flag) {
    if (e->empty) {
        // GetExpAST("") == nil: the caller leaves Item.Score untouched; evaluate to 0 like
        // ExprASTResult on a nil tree would
        PG_HIP(hipMemsetAsync(d_out, 0, (size_t)n_items * 8, ctx->stream));
    } else {
        ExprDev dev;
        dev.n = (uint32_t)e->prog.size();
        for (size_t i = 0; i < e->prog.size(); ++i) dev.prog[i] = e->prog[i];
        expr_eval_kernel<<<(n_items + 255) / 256, 256, 0, ctx->stream>>>(dev, d_vars, n_items, d_out, d_err, items_per_flag);
        PG_HIP(hipGetLastError());
    }
    return PG_OK;
}

// the fusion stage's view of a RankScore's rewrites (pipeline.hip)
int expr_num_rewrites(const pg_expr* e) { return e ? (int)e->rewrites.size() : 0; }
const char* expr_rewrite_source(const pg_expr* e, int r) { return e->rewrites[(size_t)r].source.c_str(); }
int expr_rewrite_num_vars(const pg_expr* e, int r) { return (int)e->rewrites[(size_t)r].vars.size(); }
const char* expr_rewrite_var_name(const pg_expr* e, int r, int i) { return e->rewrites[(size_t)r].vars[(size_t)i].c_str(); }
int expr_rewrite_eval_enqueue_locked(pg_ctx* ctx, const pg_expr* e, int r, const double* d_vars, uint32_t n_items, double* d_out,
                                     uint32_t* d_err, uint32_t items_per_flag) {
    const pg_expr::Rewrite& rw = e->rewrites[(size_t)r];
    if (rw.failed || rw.prog.empty()) {
        PG_HIP(hipMemsetAsync(d_out, 0, (size_t)n_items * 8, ctx->stream));
        return PG_OK;
    }
    ExprDev dev;
    dev.n = (uint32_t)rw.prog.size();
    for (size_t i = 0; i < rw.prog.size(); ++i) dev.prog[i] = rw.prog[i];
    expr_eval_kernel<<<(n_items + 255) / 256, 256, 0, ctx->stream>>>(dev, d_vars, n_items, d_out, d_err, items_per_flag);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

void set_expr_arith_error(const pg_expr* e) {
    set_error("pg_expr_eval: violation of arithmetic specification: a division by zero in '%s' "
              "(the reference panics in ExprASTResult, utils/ast/ast.go:243-249)", e->source.c_str());
}

static int expr_eval_locked(pg_ctx* ctx, const pg_expr* e, const double* d_vars, uint32_t n_items,
                            double* d_out) {
    void* p;
    int rc;
    if ((rc = scratch_reserve(ctx, 4, 4096, &p))) return rc;
    uint32_t* d_err = (uint32_t*)p + 320;
    PG_HIP(hipMemsetAsync(d_err, 0, 4, ctx->stream));
    if ((rc = expr_eval_enqueue_locked(ctx, e, d_vars, n_items, d_out, d_err, 0))) return rc;
    PG_HIP(hipMemcpyAsync(ctx->h_status + 320, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->h_status[320] != 0) {
        set_expr_arith_error(e);
        return PG_ERR_ARITH;
    }
    return PG_OK;
}

// ---- an expression over feature columns (pg_features_eval_dev) ----------------------------------------------------
// vars[v][i] = column v at rows[i] as fp64 (the evaluator's layout: one plane per variable; the column default for a row outside the store); up to 16 columns as kernel
// arguments: nothing is staged, nothing synchronises before the evaluation
struct ExprCol {
    const void* base;
    int32_t dtype;
    int32_t pad;
    double def;
};
struct ExprCols16 { ExprCol c[16]; };
__global__ void features_bind_f64_kernel(ExprCols16 cols, uint32_t nv, uint64_t rows, const uint32_t* __restrict__ cand, uint32_t n,
                                         double* __restrict__ vars) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * nv) return;
    const uint32_t v = t / n, i = t % n;
    const ExprCol c = cols.c[v];
    const uint32_t row = cand[i];
    double x = c.def;
    if (row < rows) {
        switch (c.dtype) {
            case PG_F_I32: x = (double)((const int32_t*)c.base)[row]; break;
            case PG_F_I64: x = (double)((const int64_t*)c.base)[row]; break;
            case PG_F_F32: x = (double)((const float*)c.base)[row]; break;
            default: x = ((const double*)c.base)[row];
        }
    }
    vars[t] = x;
}

}  // namespace pg

extern "C" {

int pg_features_eval_dev(pg_ctx* ctx, const pg_features* fs, const pg_expr* e, const uint32_t* d_rows, uint32_t n, double* d_out) {
    PG_REQUIRE(ctx && fs && e && d_out, "pg_features_eval_dev: NULL argument");
    if (n == 0) return PG_OK;
    PG_REQUIRE(d_rows, "pg_features_eval_dev: NULL argument");
    const size_t nv = e->vars.size();
    PG_REQUIRE(nv <= 16, "pg_features_eval_dev: %zu variables (at most 16 columns per expression)", nv);
    PG_REQUIRE((uint64_t)n * (nv ? nv : 1) < 0xFFFFFFFFull, "pg_features_eval_dev: n x variables too large");
    pg::ExprCols16 h;
    memset(&h, 0, sizeof h);
    for (size_t v = 0; v < nv; ++v) {
        const pg_features::Column* col = nullptr;
        for (const auto& c : fs->cols)
            if (c.name == e->vars[v]) { col = &c; break; }
        if (!col) {
            pg::set_error("pg_features_eval_dev: variable \"%s\" of `%s` is not a column of the feature store", e->vars[v].c_str(), e->source.c_str());
            return PG_ERR_INVALID;
        }
        h.c[v] = pg::ExprCol{col->d, col->dtype, 0, col->def};
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    double* d_vars = nullptr;
    if (nv) {
        void* buf;
        int rc;
        if ((rc = pg::scratch_reserve(ctx, 16, (size_t)n * nv * 8, &buf))) return rc;
        d_vars = (double*)buf;
        const uint32_t total = n * (uint32_t)nv;
        pg::features_bind_f64_kernel<<<(total + 255) / 256, 256, 0, ctx->stream>>>(h, (uint32_t)nv, fs->rows, d_rows, n, d_vars);
        PG_HIP(hipGetLastError());
    }
    return pg::expr_eval_locked(ctx, e, d_vars, n, d_out);
}

int pg_expr_compile(const char* source, pg_expr** out) {
    PG_REQUIRE(source && out, "pg_expr_compile: NULL argument");
    pg_expr* e = new pg_expr();
    e->source = source;
    if (e->source.empty()) {
        e->empty = true;
        *out = e;
        return PG_OK;
    }
    std::vector<pg::Token> toks;
    int rc = pg::tokenize(e->source, &toks);
    if (rc) {
        delete e;
        return rc;
    }
    if (toks.empty()) {
        // NewAST records "empty token"; ParseExpression then dereferences a nil currTok (panic)
        pg::set_error("pg_expr_compile: empty token stream for '%s'", source);
        delete e;
        return PG_ERR_PARSE;
    }
    // (the program is bounded below — kMaxProg operations — and parser and emitter recurse on the C++ stack: a source of a
    //  hundred thousand tokens must be refused here, not overflow that stack; the reference's expressions are ~10 nodes)
    if (toks.size() > 4 * (size_t)pg::kMaxProg) {
        pg::set_error("pg_expr_compile: expression too large (%zu tokens)", toks.size());
        delete e;
        return PG_ERR_UNSUPPORTED;
    }
    pg::Parser p(toks);
    const int root = p.parse_expression();
    int depth = 0;
    rc = pg::emit(p, root, e, 0, &depth);
    if (rc == PG_OK && ((int)e->prog.size() > pg::kMaxProg || depth > pg::kMaxStack)) {
        pg::set_error("pg_expr_compile: expression too large (%zu ops, depth %d)", e->prog.size(), depth);
        rc = PG_ERR_UNSUPPORTED;
    }
    if (rc) {
        delete e;
        return rc;
    }
    e->max_depth = depth;
    *out = e;
    return PG_OK;
}

// ---- the antlr subset (see the file header) -------------------------------------------------------------------------
namespace pg {
namespace {
struct AntlrParser {
    const std::string& s;
    size_t i = 0;
    pg_expr* e;
    int depth = 0, max_depth = 0, nest = 0;
    std::string err;
    AntlrParser(const std::string& src, pg_expr* out) : s(src), e(out) {}
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i; }
    bool fail(const std::string& m) { if (err.empty()) err = m; return false; }
    void push(uint32_t op, uint32_t arg, double val) {
        e->prog.push_back({op, arg, val});
        if (op == OP_CONST || op == OP_VAR) max_depth = std::max(max_depth, ++depth);
        else if (op != OP_NEG) --depth;
    }
    uint32_t var(const std::string& name) {
        uint32_t idx = 0;
        for (; idx < e->vars.size(); ++idx)
            if (e->vars[idx] == name) break;
        if (idx == e->vars.size()) e->vars.push_back(name);
        return idx;
    }
    bool param(std::string* name) {                  // ${name}
        if (s.compare(i, 2, "${") != 0) return fail("expected ${name}");
        const size_t close = s.find('}', i + 2);
        if (close == std::string::npos || close == i + 2) return fail("unterminated or empty ${…}");
        *name = s.substr(i + 2, close - i - 2);
        i = close + 1;
        return true;
    }
    bool primary() {
        ws();
        if (i >= s.size()) return fail("unexpected end of the expression");
        if (++nest > 64) return fail("nesting deeper than 64");
        bool ok = primary_inner();
        --nest;
        return ok;
    }
    bool primary_inner() {
        const char c = s[i];
        if (c == '(') {
            ++i;
            if (!expr()) return false;
            ws();
            if (i >= s.size() || s[i] != ')') return fail("missing ')'");
            ++i;
            return true;
        }
        if (c == '-') {
            // prefix minus against ^: never combined in the reference's tests, and govaluate-style grammars bind the prefix tighter
            // than the exponent (-2^2 = 4) — refused rather than evaluated on an assumption: -(a^b) or (-a)^b
            ++i;
            if (!primary()) return false;
            ws();
            if (i < s.size() && s[i] == '^')
                return fail("-a ^ b: the precedence of a prefix minus against ^ in the reference's antlr grammar is not pinned by its tests; parenthesise");
            push(OP_NEG, 0, 0.0);
            return true;
        }
        if (c == '$') {
            std::string name;
            if (!param(&name)) return false;
            push(OP_VAR, var(name), 0.0);
            return true;
        }
        if ((c >= '0' && c <= '9') || c == '.') {
            char* end = nullptr;
            const double v = strtod(s.c_str() + i, &end);
            const size_t used = (size_t)(end - (s.c_str() + i));
            if (used == 0) return fail("malformed number");
            // (strtod also reads hex floats, "inf" and "nan": not numbers of the subset)
            for (size_t j = i; j < i + used; ++j) {
                const char d = s[j];
                if (!((d >= '0' && d <= '9') || d == '.' || d == 'e' || d == 'E' || ((d == '+' || d == '-') && j > i &&
                      (s[j - 1] == 'e' || s[j - 1] == 'E'))))
                    return fail("malformed number");
            }
            i += used;
            push(OP_CONST, 0, v);
            return true;
        }
        if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_') {
            size_t j = i;
            while (j < s.size() && ((s[j] >= 'a' && s[j] <= 'z') || (s[j] >= 'A' && s[j] <= 'Z') || (s[j] >= '0' && s[j] <= '9') || s[j] == '_')) ++j;
            const std::string fn = s.substr(i, j - i);
            if (fn != "maxIndex" && fn != "maxValue")
                return fail("\"" + fn + "\" is not in the served subset (functions: maxIndex, maxValue over a ${list} property)");
            i = j;
            ws();
            if (i >= s.size() || s[i] != '(') return fail(fn + ": expected '('");
            ++i;
            ws();
            std::string name;
            if (!param(&name)) return fail(fn + ": the argument must be one ${list} property");
            ws();
            if (i >= s.size() || s[i] != ')') return fail(fn + ": the argument must be one ${list} property");
            ++i;
            // a derived variable: the host resolves "maxIndex(name)" / "maxValue(name)" from the item's list property
            push(OP_VAR, var(fn + "(" + name + ")"), 0.0);
            return true;
        }
        return fail(std::string("'") + c + "' is not in the served subset");
    }
    bool power() {
        if (!primary()) return false;
        ws();
        if (i < s.size() && s[i] == '^') {
            ++i;
            if (!primary()) return false;
            push(OP_POW, 0, 0.0);
            ws();
            if (i < s.size() && s[i] == '^') return fail("a ^ b ^ c: the associativity of the reference's antlr grammar is not pinned by its tests; parenthesise");
        }
        return true;
    }
    bool term() {
        if (!power()) return false;
        for (;;) {
            ws();
            if (i >= s.size() || (s[i] != '*' && s[i] != '/')) return true;
            if (s.compare(i, 2, "**") == 0) return fail("'**' is not in the served subset");
            const char op = s[i++];
            if (!power()) return false;
            push(op == '*' ? OP_MUL : OP_DIVF, 0, 0.0);
        }
    }
    bool expr() {
        if (!term()) return false;
        for (;;) {
            ws();
            if (i >= s.size() || (s[i] != '+' && s[i] != '-')) return true;
            const char op = s[i++];
            if (!term()) return false;
            push(op == '+' ? OP_ADD : OP_SUB, 0, 0.0);
        }
    }
};
}  // namespace
}  // namespace pg

int pg_expr_compile_typed(const char* source, const char* ast_type, pg_expr** out) {
    // GetExpASTWithType (ast.go:338-343): exactly "antlr" selects the other evaluator, anything else the default one
    if (!ast_type || strcmp(ast_type, "antlr") != 0) return pg_expr_compile(source, out);
    PG_REQUIRE(source && out, "pg_expr_compile_typed: NULL argument");
    pg_expr* e = new pg_expr();
    e->source = source;
    e->antlr = true;
    if (e->source.empty()) {                                 // GetExpASTByAntlr(""): nil
        e->empty = true;
        *out = e;
        return PG_OK;
    }
    if (e->source.size() > 16384) {
        pg::set_error("pg_expr_compile_typed: expression too large (%zu bytes)", e->source.size());
        delete e;
        return PG_ERR_UNSUPPORTED;
    }
    pg::AntlrParser p(e->source, e);
    bool ok = p.expr();
    if (ok) {
        p.ws();
        if (p.i != e->source.size()) ok = p.fail(std::string("'") + e->source[p.i] + "' is not in the served subset");
    }
    if (!ok) {
        pg::set_error("pg_expr_compile_typed: ASTType \"antlr\": %s at byte %zu of '%s' — the engine serves the subset of the "
                      "go-antlr-valuate language the reference's tests pin (+ - * / ^, parentheses, ${name}, maxIndex / maxValue; "
                      "utils/ast/ast_test.go:30-56,90-167,213-300) and refuses the rest rather than evaluate it differently",
                      p.err.c_str(), p.i, source);
        delete e;
        return PG_ERR_UNSUPPORTED;
    }
    if (e->prog.size() > (size_t)pg::kMaxProg || p.max_depth > pg::kMaxStack) {
        pg::set_error("pg_expr_compile_typed: expression too large (%zu operations, depth %d)", e->prog.size(), p.max_depth);
        delete e;
        return PG_ERR_UNSUPPORTED;
    }
    e->max_depth = p.max_depth;
    *out = e;
    return PG_OK;
}

int pg_expr_is_antlr(const pg_expr* e) { return e && e->antlr ? 1 : 0; }

int pg_expr_free(pg_expr* e) {
    delete e;
    return PG_OK;
}

int pg_expr_num_vars(const pg_expr* e) { return e ? (int)e->vars.size() : 0; }
}  // extern "C"
namespace pg {
void expr_hold(const pg_expr* e) { e->holders.fetch_add(1, std::memory_order_acq_rel); }
void expr_release(const pg_expr* e) { e->holders.fetch_sub(1, std::memory_order_acq_rel); }
}  // namespace pg
extern "C" {

int pg_expr_set_score_rewrites(pg_expr* rank_score, uint32_t n, const char* const* sources, const pg_expr* const* exprs) {
    PG_REQUIRE(rank_score && (n == 0 || (sources && exprs)), "pg_expr_set_score_rewrites: NULL argument");
    PG_REQUIRE(n <= (uint32_t)pg::kMaxRewrites, "pg_expr_set_score_rewrites: %u rewrites (at most %d)", n, pg::kMaxRewrites);
    if (rank_score->holders.load(std::memory_order_acquire) != 0) {
        pg::set_error("pg_expr_set_score_rewrites: the expression is bound by a live pipeline (a coalescer, or a batch that has not been "
                      "ended): attach the rewrites before creating any pipeline from it");
        return PG_ERR_INVALID;
    }
    PG_REQUIRE(n == 0 || !rank_score->empty, "pg_expr_set_score_rewrites: the reference rewrites scores only in front of a RankScore "
               "(service/rank/rank_service.go:339-353); this one is empty");
    std::vector<pg_expr::Rewrite> rw(n);
    for (uint32_t i = 0; i < n; ++i) {
        PG_REQUIRE(sources[i] && sources[i][0], "pg_expr_set_score_rewrites: rewrite %u has no source name", i);
        for (uint32_t j = 0; j < i; ++j)
            PG_REQUIRE(rw[j].source != sources[i], "pg_expr_set_score_rewrites: source \"%s\" twice (ScoreRewrite is a map)", sources[i]);
        rw[i].source = sources[i];
        if (!exprs[i]) {
            rw[i].failed = true;
            continue;
        }
        PG_REQUIRE(exprs[i]->rewrites.empty(), "pg_expr_set_score_rewrites: \"%s\": a rewrite expression cannot carry rewrites itself", sources[i]);
        if (exprs[i]->empty) continue;                  // GetExpAST(""): a nil tree evaluates to 0
        rw[i].prog = exprs[i]->prog;
        rw[i].vars = exprs[i]->vars;
    }
    rank_score->rewrites.swap(rw);
    return PG_OK;
}

const char* pg_expr_var_name(const pg_expr* e, int i) {
    if (!e || i < 0 || i >= (int)e->vars.size()) return "";
    return e->vars[i].c_str();
}

int pg_expr_eval_dev(pg_ctx* ctx, const pg_expr* e, const double* d_vars, uint32_t n_items,
                     double* d_out_scores) {
    PG_REQUIRE(ctx && e && d_out_scores, "pg_expr_eval_dev: NULL argument");
    PG_REQUIRE(e->vars.empty() || d_vars, "pg_expr_eval_dev: vars is NULL but the expression has parameters");
    if (n_items == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::expr_eval_locked(ctx, e, d_vars, n_items, d_out_scores);
}

int pg_expr_eval(pg_ctx* ctx, const pg_expr* e, const double* vars, uint32_t n_items, double* out_scores) {
    PG_REQUIRE(ctx && e && out_scores, "pg_expr_eval: NULL argument");
    PG_REQUIRE(e->vars.empty() || vars, "pg_expr_eval: vars is NULL but the expression has parameters");
    if (n_items == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    void* buf;
    int rc;
    const size_t vb = e->vars.size() * (size_t)n_items * 8, ob = (size_t)n_items * 8;
    if ((rc = pg::scratch_reserve(ctx, 5, vb + ob + 256, &buf))) return rc;
    double* d_v = (double*)buf;
    double* d_o = (double*)((char*)buf + ((vb + 255) & ~(size_t)255));
    if (vb) PG_HIP(hipMemcpyAsync(d_v, vars, vb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::expr_eval_locked(ctx, e, d_v, n_items, d_o))) return rc;     // scores untouched on error
    PG_HIP(hipMemcpyAsync(out_scores, d_o, ob, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

}  // extern "C"
