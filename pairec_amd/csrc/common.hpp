// common.hpp — shared host-side plumbing of libpairec_gpu.so (context, error convention).
// gfx950 (MI355X) only: wave = 64 lanes, 256 CUs, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/pairec_gpu.h"

namespace pg {

struct PipeRun;          // pipeline.hpp
void set_error(const char* fmt, ...);

#define PG_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::pg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,  \
                            __LINE__);                                                         \
            return PG_ERR_DEVICE;                                                              \
        }                                                                                      \
    } while (0)

#define PG_REQUIRE(cond, ...)                    \
    do {                                         \
        if (!(cond)) {                           \
            ::pg::set_error(__VA_ARGS__);        \
            return PG_ERR_INVALID;               \
        }                                        \
    } while (0)

constexpr int kWave = 64;
constexpr int kMaxQueriesExact = 64;   // exact fp32-MFMA scan: one or two 32-query column blocks
constexpr int kMaxQueries = 256;       // screened scan (int8 or bf16 filter + exact rescoring): up to eight blocks

// Scratch arena: grows on demand, never shrinks; owned by the context, used by one call at a time
// (calls on a context are serialised by ctx->mu).
struct Scratch {
    void* p = nullptr;
    size_t cap = 0;
};

// Developer / tuning knobs.  Read from the environment ONCE, in pg_init (a serving process must not change
// plans because of a stray variable later on, and getenv races with a host application's setenv); tests and
// A/B scripts change them per context with pg_set_option.
struct Knobs {
    uint32_t screen_min = 0;       // PG_SCREEN_MIN: batches up to this many queries use the exact scan
    bool recall_exact = false;     // PG_RECALL_EXACT: never screen
    double pilot_fraction = 0.0;   // PG_PILOT_FRACTION: sample fraction of the pilot plan (0 = default 1/64)
    bool no_pilot = false;         // PG_NO_PILOT: skip the pilot plan
    double pilot_sigmas = 6.0;     // PG_PILOT_SIGMAS: K' = m + sigmas sqrt(m) + 8 (margin of the pilot threshold)
    double chunk_growth = 0.0;     // PG_CHUNK_GROWTH: geometric growth of the grow plan (0 = default)
    uint32_t seed_rows = 8192;     // PG_SEED_ROWS: exact seed of the pilot sample
    double pilot_growth = 0.0;     // PG_PILOT_GROWTH: > 0 = geometric chunks over the sample instead of seed + one launch
    bool debug_scan = false;       // PG_DEBUG_SCAN: print per-launch times and suspect counts
    bool screen_bf16 = false;      // PG_SCREEN_BF16: bf16 shadow for dim-128 tables too
    bool screen_i8 = false;        // PG_SCREEN_I8: int8 shadow even for heavy-tailed tables
    bool no_refine = false;        // PG_NO_REFINE: the pilot plan's full pass keeps the sample's threshold throughout
    uint32_t refine_min_rows = 1u << 24;   // PG_REFINE_MIN_ROWS: smallest table whose full pass is split for the refinement
    bool no_screen_i4 = false;     // PG_NO_SCREEN_I4: small batches stay on the int8 screen
    bool no_screen_i4m = false;    // PG_NO_SCREEN_I4M: batches of 5..64 queries stay on the int8 screen
    uint32_t i4m_min_queries = 1;  // PG_I4M_MIN_QUERIES: smallest batch it serves (below, and for squared-Euclidean recalls and tables without an int8 shadow: recall_i4.hip's vector-ALU screen, exact re-scoring of all its suspects; one query: 1.08 against 1.17 ms per 100 M rows)
    uint32_t i4m_max_queries = 64; // PG_I4M_MAX_QUERIES: largest batch the 4-bit matrix-pipe screen serves (<= kI4mMaxQueries)
    double i4m_max_lambda = 2.2;   // PG_I4M_MAX_LAMBDA: largest pg_table::lam4 it is used for
    double i4m_max_pairs = 2.4e7;  // PG_I4M_MAX_PAIRS: ... and the most (row, query) pairs per pass its 4-bit stage may be expected to pass on
    uint32_t i4_min_rows = 1u << 22; // PG_I4_MIN_ROWS: smallest table the 4-bit screen is built for
    double i4_max_lambda = 1.7;    // PG_I4_MAX_LAMBDA: largest pg_table::lam4 the 4-bit screen is used for
    bool rank_no_ws = false;       // PG_RANK_NO_WS: streaming DNN3 kernel instead of the weights-stationary one
    uint32_t split_sort_max = 96;  // PG_SPLIT_SORT_MAX: up to this many lists (of 1025 … 8192 items, ~450 K items in all) per call are sorted run by run over the chip (split_sort.hpp; 0 = never)
    uint32_t rank_sort_max = 32;   // PG_RANK_SORT_MAX: up to this many lists per call are sorted by counting ranks (0 = never) ...
    double rank_sort_work = 7e7;   // PG_RANK_SORT_WORK: ... while lists x items^2 stays under this (lists of <= 8192 items; the top-K's final order, one compare per key, takes 1.5 x)
    bool sort_lds = false;         // PG_SORT_LDS: LDS bitonic sort instead of the register-resident one
    bool dpp_valu = false;         // PG_DPP_VALU: the DPP kernel matrix on the fp64 vector pipe (round-4 kernel) instead of the fp64 matrix pipe (A/B; same bits)
    bool fm2t_irs = false;         // PG_FM2T_IRS: cfg 4's item-record rank on the producer / consumer kernel (rank_ir.hip) instead of rank_is.hip (A/B)
    bool no_r2 = false;            // PG_NO_R2: never refine the int8 screen's suspects on the residual shadow
    double r2_min_factor = 3.0;    // PG_R2_MIN_FACTOR: ... refine once a table's passes average more than this many suspects per answer (x K)
    uint32_t max_rec_scale = 16;   // PG_MAX_REC_SCALE: the 256-query pass's hit-record areas grow up to this many times their default size with a table
                                   // whose batches overflow them (16: 80 B x 123 M records = 9.8 GB per context at K = 5 000); 1 = never (exact scan instead)
    double coalescer_rejoin_us_per_caller = 2.0;   // PG_COALESCER_REJOIN_US: the hold lasts at most 50 + this x n microseconds for a batch of n
    bool coalescer_rejoin = true;  // PG_COALESCER_NO_REJOIN clears it: after a completion a waiting partial batch is held for the callers just answered
    bool no_predict = false;       // PG_NO_PREDICT: never replace the pilot sample by the learned threshold model
    double predict_max_factor = 4.0;   // PG_PREDICT_MAX_FACTOR: a table whose predicted thresholds admit more than this many candidates per answer goes back to the pilot plan
    double predict_sigmas = 4.5;   // PG_PREDICT_SIGMAS: margin of the predicted threshold, in standard deviations of the observed quantile
    uint32_t predict_min_rows = 1u << 22;   // PG_PREDICT_MIN_ROWS: smaller tables are launch-bound either way
    uint32_t screen_early_share_narrow = 512;   // PG_SCREEN_EARLY_SHARE_NARROW: the same for the 8-wave kernels of <= 128 queries
    bool l2_exact = false;                  // PG_L2_EXACT: squared-Euclidean recalls always on the exact scan (A/B runs)
    uint32_t where_compact_max_rows = 8u << 20;   // PG_WHERE_COMPACT_MAX_ROWS: a filter admitting at most this many rows (half of it for <= 4 queries) ...
    uint32_t where_compact_min_ratio = 8;         // PG_WHERE_COMPACT_MIN_RATIO: ... and at most 1/ratio of the table is served from a compact copy of them
    double l2_max_slack = 1.0;              // PG_L2_MAX_SLACK: largest pg_table::l2_slack the per-BLOCK cutoff is used for (above: the per-row test)
    uint32_t screen_early_share = 604;      // PG_SCREEN_EARLY_SHARE: share (x 1024) of a SIMD's blocks given to its older wave (256-query screen)
};

}  // namespace pg

struct pg_table {
    // Readers — every enqueue that dereferences the table's pointers, for as long as it reads them — share this lock;
    // pg_table_swap / _upload / _fill take it exclusively and drain the device first.  A request batch (recall, rank,
    // DPP gather: one enqueue) therefore sees ONE table version even when a second context (a coalescer's sibling)
    // enqueues while the table is being swapped.  Lock order: ctx->mu, then the table.
    mutable std::shared_mutex rw;
    // bumped by every exclusive section (upload, fill, swap): a recall job remembers the value it was prepared against, and a
    // re-plan or a patch that finds another one restarts the whole batch on the new rows instead of mixing generations
    mutable std::atomic<uint64_t> generation{0};
    float* d = nullptr;          // [rows][dim] fp32 row-major
    uint64_t rows = 0;
    uint32_t dim = 0;
    uint64_t row_offset = 0;     // global row id of local row 0 (sharded tables)
    // a filtered view (pg_table_view_create): the rows of a source table that a WhereClause admits, in row order; recalls
    // report d_row_map[local row] + map_offset — the source's row ids
    uint32_t* d_row_map = nullptr;
    uint64_t map_offset = 0;
    // lazily computed for the screened recall (invalidated by upload / fill): statistics and the shadow of
    // the rows that the screen streams instead of the fp32 rows (the exact re-scoring still gathers fp32):
    //   dim 128: int8, X = rint(x / s8) with ONE scale s8 = max|x| / 127 for the table, [rows + 64][dim] bytes
    //            — a quarter of the fp32 bytes; resid8 = max over rows of ||x - s8 X||_2 (measured, not assumed)
    //   dim 64, and dim-128 tables whose value range defeats one int8 scale (heavy tails, outliers): bf16 (RNE of
    //            every fp32 value), [rows + 64][dim] — half the bytes — plus the largest row norm of every 32-row block
    bool stats_valid = false;
    bool all_finite = false;
    float max_norm = 0.0f;       // upper bound of the rows' L2 norms
    uint16_t* d16 = nullptr;     // bf16 shadow; allocated on first use, kept across rebuilds
    float* dnorm2 = nullptr;     // with the bf16 shadow: largest row norm^2 of every 32-row block (the bf16 bound is relative)
    int8_t* d8 = nullptr;        // int8 shadow; likewise
    bool shadow_is_i8 = false;   // which of the two the current statistics belong to
    float s8 = 0.0f;             // int8 scale
    float resid8 = 0.0f;         // upper bound of the rows' quantisation residual (L2)
    bool shadow_failed = false;  // allocation failed once: stay on the exact scan
    float* d_nx = nullptr;       // squared-Euclidean recall: |x|^2 of every row [rows + 64] (lazily, invalidated by upload / fill)
    float* d_nxmin = nullptr;    // ... and the smallest of every 32-row block (the int8 screen's per-block cutoff under that metric)
    bool nx_valid = false;
    float l2_slack = 0.0f;       // mean (|x|^2 - the block's smallest) / 2 in units of the score spread: above Knobs::l2_max_slack the exact scan serves that metric
    // recall_i4.hip: the 4-bit shadow that the full pass of a small batch streams (dim 128, built on the first such
    // recall, 68 B per row): nibbles [rows + 64][64 B], one fp32 scale per row, and the bound's measured constants
    uint8_t* d4 = nullptr;
    uint32_t* d4s = nullptr;     // row scale (bf16, low half) | row residual bound (bf16, high half)
    bool i4_ok = false, i4_failed = false;
    float rho4 = 0.0f;           // max over rows of ||x - x^|| / (s_row sqrt(dim)) (diagnostic)
    float rmax4 = 0.0f;          // max over rows of ||x - x^|| (upper bound)
    float lam4 = 0.0f;           // mean residual term in units of the score spread (decides whether the shadow pays)
    float i4m_pairs = 0.0f;      // recall_i4m.hip: running average of the (row, query) pairs its 4-bit stage lets through, per query
    // recall_r2.hip: the int8 RESIDUAL shadow (x = s8 X8 + s8r Xr + e2) that the refinement stage of crowded tables gathers beside
    // the int8 shadow; built the first time the table shows more than Knobs::r2_min_factor screen suspects per answer
    int8_t* d8r = nullptr;
    float s8r = 0.0f, resid2 = 0.0f;   // its scale (s8 / 254), the measured upper bound of ||e2||
    bool r2_ok = false, r2_failed = false;
    float wide_susp = 0.0f;      // running average of the int8 screen's suspects per query (passes without a 4-bit stage)
    uint32_t rec_scale = 0;      // recall.hip: the 256-query pass's hit-record areas, x their default size (0 = 1; doubled when a batch overflows them)
    uint32_t prefix_failures = 0; // batches whose refined thresholds failed verification for most queries (ordered rows): two → no refinement
    // Threshold predictor (recall.hip, DESIGN.md 4.1, plan 0): a Gaussian model of a query's scores over the rows — mean
    // vector and covariance from a row sample, built with the statistics — and the quantile z = (K-th best score −
    // mean) / sigma actually observed for the batches served so far.  Once the observed z is tight, a batch's first
    // thresholds come from the model instead of a pilot sample.  A hint only: every plan is verified, results are exact.
    float* d_pred = nullptr;      // [dim] mean | [dim][dim] covariance | 4 doubles: n, sum z, sum z^2, min z
    bool pred_model = false;      // mean / covariance belong to the current rows
    uint32_t pred_k = 0;          // the K the observations belong to
    double pred_n = 0.0, pred_sum = 0.0, pred_sum2 = 0.0, pred_min = 0.0, pred_total = 0.0;   // host copy as of the last verified batch
    uint32_t pred_backoff = 0;    // batches that stay on the pilot plan after a prediction failed verification
    uint32_t pred_failures = 0;
    // screened plans that ended in an overflowing suspect / record / candidate list (rows crowded within the screen's error of
    // every query's K-th score: DESIGN.md 4.1): after two in a row the table's recalls start on the exact scan for a while
    uint32_t screen_overflow_streak = 0, screen_backoff = 0;
};

// rank model weights resident in HBM (rank_mlp.hip loads them)
struct pg_model {
    pg_model_kind kind;
    int prec;
    uint32_t d_user = 0, d_item = 0, h1 = 0, h2 = 0;           // DNN3
    uint32_t nuf = 0, nif = 0, k = 0, th = 0, to = 0, vocab = 0;  // two-tower
    float b3 = 0.f, fm_b = 0.f;
    uint32_t n_out = 1;     // DNN3: heads on the shared trunk (PG_MODEL_DNN3_MULTI; kind is stored as PG_MODEL_DNN3)
    float* b3v = nullptr;   // device [n_out] head biases (b3 = b3v[0])
    // device buffers
    float* w1u = nullptr;   // [d_user][h1] (DNN3) / uw1 (two-tower), operand-rounded fp32
    float* b1 = nullptr;    // b1 / ub1
    float* uw2 = nullptr;   // two-tower user layer 2
    float* ub2 = nullptr;
    void* w1p = nullptr;    // packed item-side layer 1
    void* w2p = nullptr;    // packed layer 2
    void* w1p_lo = nullptr; // PG_PREC_BF16X3: the lo fragments (inside the w1p / w2p allocations)
    void* w2p_lo = nullptr;
    float* c1_shared = nullptr;   // two-tower: ib1
    float* b2 = nullptr;
    float* w3 = nullptr;    // DNN3 head(s): [n_out][h2]
    float* fields = nullptr;          // two-tower: all field tables, one allocation
    const float** d_field_emb = nullptr;
    const float** d_field_lin = nullptr;
    std::vector<void*> allocs;
};

// features.hip owns the columns; rank_mlp.hip reads them when it materialises item records
struct pg_features {
    uint64_t rows = 0;
    struct Column {
        std::string name;
        int dtype = 0;
        void* d = nullptr;          // [rows] of the column type
        double def = 0.0;           // default for rows outside the store
    };
    std::vector<Column> cols;
};

// materialised item side of an FM + two-tower model (rank_mlp.hip): [rows + 1][kItemRowFloats] fp32
struct pg_item_rows {
    const pg_model* m = nullptr;
    const pg_features* fs = nullptr;
    int32_t cols[16] = {0};
    uint64_t rows = 0;
    float* d = nullptr;
};

struct pg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    std::mutex mu;               // serialises calls on this context
    // named scratch slots (transient: a slot's contents are valid until the next reserve of the same slot on this context;
    // users that share a slot are stages that never overlap on the context's one stream):
    //   0 table upload staging / fm2t rows' gathered field ids     1 table / features host staging
    //   2, 3 recall.hip candidate lists and thresholds             4 small status words (expr, misc, recall, recall_i4)
    //   5 host-buffer entry points' staging (rank, recall, sort, dpp, ssd, expr)
    //   6 rank tile table + request partials   7 sort / dpp / ssd work areas   8 recommend pipeline intermediates (post_scratch)
    //   9 group.hip   10 re-rank stage (DPP candidates)   11 recall.hip: the screened pass's record regions
    //   12, 13 recall.hip   14 rank_mlp.hip: head partials of the weights-stationary multi-head kernel   15 pg_fuse_scores_dev
    //   16 pg_features_eval_dev: the bound variables
    pg::Scratch scratch[17];
    std::mutex pool_mu;          // guards pipe_free
    std::vector<pg::PipeRun*> pipe_free;     // per-batch status blocks / events of the device-resident pipelines
    std::map<const void*, size_t> dyn_lds;   // kernels whose dynamic-LDS limit was raised on this device
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // 0/1 probes, 2/3 rank, 4/5 sort
    std::vector<hipEvent_t> ev_pool;   // per-launch timing events (scan kernel roofline figure)
    uint32_t last_scan_launches = 0;
    bool rank_timing_pending = false;
    pg_stats_t stats{};
    pg::Knobs knobs;
    double last_scan_ms = 0.0;
    uint64_t last_scan_bytes = 0;
    // pinned host staging for small status words
    uint32_t* h_status = nullptr;
    bool timers_off = false;            // (under mu) the call being enqueued records no stage-timer events: TimersScope
};

namespace pg {

typedef std::shared_lock<std::shared_mutex> TableRead;
// shared access to two tables (a recall whose queries are rows of a trigger table): taken in ADDRESS order, the order
// TableWrite(a, b) uses — a reader holding t and waiting for the trigger table against a swap of the two would deadlock
struct TableRead2 {
    TableRead a, b;
    TableRead2(const pg_table* x, const pg_table* y) {
        if (y && y < x) std::swap(x, y);
        a = TableRead(x->rw);
        if (y && y != x) b = TableRead(y->rw);
    }
};
// exclusive access to one or two tables (address order) with nothing in flight on the device
struct TableWrite {
    std::unique_lock<std::shared_mutex> a, b;
    TableWrite(const pg_table* x, const pg_table* y = nullptr) {
        if (y && y < x) std::swap(x, y);
        a = std::unique_lock<std::shared_mutex>(x->rw);
        x->generation.fetch_add(1, std::memory_order_relaxed);
        if (y && y != x) {
            b = std::unique_lock<std::shared_mutex>(y->rw);
            y->generation.fetch_add(1, std::memory_order_relaxed);
        }
    }
};

// ---- recall.hip: a recall whose plans are enqueued without host synchronisation and verified later --------
struct RecallScratch {
    float* qpad;
    float* thr;
    uint32_t* cnt;
    uint32_t* overflow;
    uint64_t* cand[2];
    uint32_t cap;
    // screened scan only
    uint4* qb16;
    float* eps;
    float* thr_screen;
    uint32_t* susp_cnt;      // [kMaxQueries]
    uint32_t* susp;          // [kMaxQueries][cap] suspect rows of the current launch
    float* qscale;           // [kMaxQueries] int8 screen: the queries' scales
    float* thr_ref;          // [kMaxQueries] the thresholds the pilot plan's refinement step raised (verified after the pass)
    uint32_t* q4;            // 4-bit screen: [4][32] int8 queries + [4][4] constants (recall_i4.hip)
    float* pred_ms;          // [kMaxQueries][2] the threshold model's mean and sigma of every query's scores
    // mid-batch screen (recall_i4m.hip): the int8 queries in plain order + the 4-bit bound's constants; the suspects that
    // passed its int8 stage, [kI4mMaxQueries][cap]
    uint32_t* q4m;
    uint32_t* susp2;
    uint32_t* susp2_cnt;     // [kI4mMaxQueries]
    // refinement stage of crowded tables (recall_r2.hip): the queries in sixteen bits + constants; its survivors' counts
    // (their lists share susp2, which holds kMaxQueries lists)
    uint32_t* q16;
    uint32_t* susp2w_cnt;    // [kMaxQueries]
    uint32_t* status;        // [kRecallStatusWords] a job's status words, packed by one kernel for ONE copy to the host
    uint32_t* cnt_seen;      // [kMaxQueries] the candidates each query had collected when the last select kept K of them
};
constexpr uint32_t kRecallStatusWords = 640;
// A predicate over an integer feature column that restricts a recall's candidates (HologresVectorConf.WhereClause of the
// reference, hologres_vector_recall.go:49-62, in the one shape the device serves: `column OP constant`).  Rows that fail it
// never become candidates — it is applied where candidates are made (exact re-scoring, the exact scan's hit path), so every
// threshold the plans derive is a threshold of the FILTERED top-K and the answers stay exact.
struct RowFilter {
    const void* col = nullptr;   // [rows] int32 / int64 device column; nullptr = no filter
    int dtype = 0;               // PG_F_I32 or PG_F_I64
    int op = 0;                  // pg_where_op: 0 >, 1 >=, 2 <, 3 <=, 4 ==, 5 !=
    long long val = 0;
    long long admitted = -1;     // host side: rows the filter admits when the caller has counted them already (-1: not yet)
};
__host__ __device__ inline bool row_filter_pass(const RowFilter& f, uint32_t row) {
    if (!f.col) return true;
    const long long v = f.dtype == 2 ? reinterpret_cast<const long long*>(f.col)[row] : (long long)reinterpret_cast<const int32_t*>(f.col)[row];
    switch (f.op) {
        case 0: return v > f.val;
        case 1: return v >= f.val;
        case 2: return v < f.val;
        case 3: return v <= f.val;
        case 4: return v == f.val;
        default: return v != f.val;
    }
}

struct RecallJob {
    // set by the caller
    pg_ctx* ctx = nullptr;
    const pg_table* t = nullptr;
    const float* d_queries = nullptr;       // [nq][dim] device
    uint32_t nq = 0, k = 0;
    uint64_t* d_out_rows = nullptr;         // [nq][k] device
    float* d_out_scores = nullptr;          // [nq][k] device
    uint32_t* d_out_count = nullptr;        // [nq] device, optional
    uint32_t* h_status = nullptr;           // pinned host, >= 1 + nq words: [0] overflow flag, [1 + q] valid count of query q
    std::vector<hipEvent_t>* events = nullptr;   // timing events (grown on demand); one job at a time per pool
    bool skip_pilot = false;                // start with the growing-chunk plan (the re-run of a query the pilot failed)
    bool l2 = false;                        // rank by smallest squared Euclidean distance (scores out = distances)
    bool l2_per_row = false;                // ... its screened pass tests every row against its own norm (rows of mixed norms)
    RowFilter filter{};                     // restrict the candidates to the rows that pass (col = nullptr: none)
    uint32_t rows_qualified = 0;            // ... how many rows do (counted in recall_job_prepare)
    bool exact_only = false;                // no statistics, no shadow: the exact scan (the compact table of a selective filter)
    // state (recall_job_*)
    RecallScratch rs{};
    uint32_t* d_count = nullptr;
    uint32_t rows = 0, nblocks = 0;
    bool screen = false;
    bool screen4 = false;                   // the pilot plan's full pass streams the 4-bit shadow (nq <= kI4MaxQueries)
    bool stage2 = false;                    // the int8 screen's suspects pass the two-digit refinement (recall_r2.hip) before the exact re-scoring
    bool screen4m = false;                  // ... through the matrix pipe, suspects thinned on the int8 shadow (kI4MaxQueries < nq <= kI4mMaxQueries)
    int plans[4] = {0, 0, 0, 0};
    bool predict = false;                   // plans[0] takes its first thresholds from the table's threshold model
    bool pred_observe = false;              // the table has a model: this job contributes its observed quantiles
    bool observed = false;                  // ... and the enqueued plan did
    double z_lo = 0.0;
    bool susp_stat = false;                 // the enqueued plan reports its suspect counts with the status words
    bool timers = true;                     // this enqueue recorded its stage-timer events (false: a coalescer batch — recall_job_check reads none)
    bool stat_wide = false;                 // ... and they are the int8 screen's own (no 4-bit stage in front)
    bool refined = false;                   // the enqueued pilot plan raised its thresholds after the first quarter
    int n_plans = 0, next_plan = 0, enqueued_plan = -1;
    uint32_t stride = 1, sample_blocks = 0, k_pilot = 0, perm_mul = 1;
    uint32_t n_ev = 0;
    double scan_ms = 0.0, total_ms = 0.0;
    uint64_t scanned_rows = 0;
    uint64_t scan_bytes = 0;
    uint32_t scan_launches = 0;
    // after a failed check of the pilot plan without overflow: the queries that ended short of K candidates (their
    // sample threshold was too high).  A handful can be re-run one by one instead of re-running the whole batch.
    std::vector<uint32_t> failed;
    uint64_t table_gen = 0;                 // t->generation when the job was prepared (the statistics, shadow and plans are that version's)
};
constexpr size_t kMaxPatchQueries = 8;
// All four: caller holds ctx->mu.  prepare may synchronise once (a table's statistics / shadow on first use).
int recall_job_prepare(RecallJob* j);
int recall_job_enqueue(RecallJob* j);                 // enqueue the next plan + the status copy into h_status
int recall_job_check(RecallJob* j, bool* ok);         // after the stream passed the status copy: did the plan hold?
void recall_job_finish(RecallJob* j);                 // publish timing / counters into ctx
int recall_dev_locked(pg_ctx* ctx, const pg_table* t, const float* d_queries, uint32_t nq, uint32_t k,
                      uint64_t* d_out_rows, float* d_out_scores, uint32_t* out_count, uint32_t* d_out_count,
                      bool skip_pilot = false, bool l2 = false, const RowFilter* filter = nullptr, bool exact_only = false);
// re-run the failed queries of `j` (at most kMaxPatchQueries) one by one, synchronously, writing into their slices of
// the job's outputs and their valid counts into counts[q]; caller holds ctx->mu
int recall_patch_failed_locked(RecallJob* j, uint32_t* counts);
// Stage timers (HIP events around the recall plan, its scan launches and the rank stage: pg_stats' last_*_ms, pg_last_scan_kernel_ms)
// cost ~6 us of idle queue per record — 40-60 us of a 1.3-1.8 ms small-batch step.  A caller that reads none of them (the coalescer:
// its statistics are enqueue -> completion times) switches them off for the calls it enqueues; caller holds ctx->mu.
struct TimersScope {
    pg_ctx* ctx;
    bool prev;
    TimersScope(pg_ctx* c, bool timers) : ctx(c), prev(c->timers_off) { c->timers_off = prev || !timers; }      // (only ever off: pg_set_option "stage_timers" 0 stays in force)
    ~TimersScope() { ctx->timers_off = prev; }
};
int recall_scratch(pg_ctx* ctx, uint32_t dim, uint32_t k, RecallScratch* rs);
int launch_select(pg_ctx* ctx, uint32_t nq, const uint64_t* in, uint64_t* out, uint32_t* cnt, float* thr,
                  uint32_t cap, uint32_t k, int thr_only = 0, uint32_t* cnt_seen = nullptr);
int final_launch(pg_ctx* ctx, const uint64_t* cand, const uint32_t* cnt, uint32_t cap, uint32_t nq, uint32_t k,
                 uint64_t row_offset, uint64_t* d_out_rows, float* d_out_scores, uint32_t* d_out_count);
int ensure_table_stats(pg_ctx* ctx, const pg_table* tc);
// recall_i4.hip (caller holds ctx->mu)
constexpr uint32_t kI4MaxQueries = 4;
int ensure_table_i4(pg_ctx* ctx, const pg_table* tc);
int screen4_prep_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs);
int screen4_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs, uint32_t nq, uint32_t row_begin, uint32_t row_end,
                   uint32_t cap4, const float* l2_nqv = nullptr);
uint32_t screen4_rescore_blocks();
// recall_i4m.hip (caller holds ctx->mu): the same shadow through the int8 matrix pipe for 5 .. kI4mMaxQueries queries
constexpr uint32_t kI4mMaxQueries = 64;
constexpr uint32_t kQ4mWords = kI4mMaxQueries * 32 + kI4mMaxQueries * 4 + 4;
int screen4m_prep_launch(pg_ctx* ctx, const RecallScratch& rs, uint32_t nq);
// recall_r2.hip (caller holds ctx->mu)
int ensure_table_r2(pg_ctx* ctx, const pg_table* tc);
int rescreen16_prep_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs, uint32_t nq);
int rescreen16_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs, uint32_t nq);
int screen4m_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs, uint32_t nq, uint32_t row_begin, uint32_t row_end,
                    uint32_t cap1, bool susp2_clean = false);
int topk_merge_strided_locked(pg_ctx* ctx, const uint64_t* d_rows, const float* d_scores, uint32_t nq, uint32_t nlists,
                              uint32_t per_list, size_t row_ls, size_t row_qs, size_t sc_ls, size_t sc_qs, uint32_t k,
                              uint64_t* d_out_rows, float* d_out_scores, uint32_t* d_out_count);
int topk_merge_locked(pg_ctx* ctx, const uint64_t* d_rows, const float* d_scores, uint32_t nq, uint32_t nlists,
                      uint32_t per_list, int list_major, uint32_t k, uint64_t* d_out_rows, float* d_out_scores,
                      uint32_t* d_out_count);

// ---- stage launchers shared with pipeline.hip (caller holds ctx->mu; nothing synchronises) -------------------
// head o of a multi-head model writes d_out + o * out_stride (out_stride = 0: n_items)
int rank_dnn3_dev_locked(pg_ctx* ctx, const pg_model* m, const pg_table* t, const float* d_user,
                         const uint32_t* d_cand, const uint32_t* d_off, uint32_t n_req, uint32_t n_items,
                         float* d_out, size_t out_stride = 0);
int fm2t_user_embedding_locked(pg_ctx* ctx, const pg_model* m, const float* d_user, uint32_t n_req, float* d_out);
int rank_fm2t_irows_dev_locked(pg_ctx* ctx, const pg_model* m, const pg_item_rows* ir, const float* d_user, const int32_t* d_ufids,
                               const uint32_t* d_cand, const uint32_t* d_off, uint32_t n_req, uint32_t n_items, float* d_out);
int rank_fm2t_rows_dev_locked(pg_ctx* ctx, const pg_model* m, const pg_features* fs, const int32_t* item_field_cols,
                              const float* d_user, const int32_t* d_ufids, const uint32_t* d_cand, const uint32_t* d_off,
                              uint32_t n_req, uint32_t n_items, float* d_out);
// d_err: device flags, OR-ed with 1 where an item divides by zero; item i reports into d_err[i / items_per_flag]
// (items_per_flag = 0: one flag for the call)
int expr_eval_enqueue_locked(pg_ctx* ctx, const pg_expr* e, const double* d_vars, uint32_t n_items, double* d_out,
                             uint32_t* d_err, uint32_t items_per_flag);
void set_expr_arith_error(const pg_expr* e);
// RankConfig.ScoreRewrite attached to a RankScore expression (pg_expr_set_score_rewrites; expr.hip)
constexpr int kMaxRewrites = 8;
// Holders of a compiled RankScore: everything that keeps a variable binding made from it (recommend_bind_vars: coalescers for their
// lifetime, tickets until they are ended, pg_fuse_scores_dev for the call).  pg_expr_set_score_rewrites refuses while there is
// one — the bindings are sized for the rewrites attached at that moment.
void expr_hold(const pg_expr* e);
void expr_release(const pg_expr* e);
struct ExprHold {
    const pg_expr* e = nullptr;
    ExprHold() = default;
    ExprHold(const ExprHold&) = delete;
    ExprHold& operator=(const ExprHold&) = delete;
    void take(const pg_expr* x) {
        if (e) expr_release(e);
        e = x;
        if (e) expr_hold(e);
    }
    ~ExprHold() { if (e) expr_release(e); }
};
int expr_num_rewrites(const pg_expr* e);
const char* expr_rewrite_source(const pg_expr* e, int r);
int expr_rewrite_num_vars(const pg_expr* e, int r);
const char* expr_rewrite_var_name(const pg_expr* e, int r, int i);
int expr_rewrite_eval_enqueue_locked(pg_ctx* ctx, const pg_expr* e, int r, const double* d_vars, uint32_t n_items, double* d_out,
                                     uint32_t* d_err, uint32_t items_per_flag);
int table_gather_locked(pg_ctx* ctx, const pg_table* t, const uint32_t* d_rows, uint32_t n, float* d_out);
int dpp_run_locked(pg_ctx* ctx, const float* d_emb32, const double* d_hook, const double* d_rel, uint32_t R, uint32_t n,
                   uint32_t d, uint32_t hook_dim, double alpha, uint32_t topn, uint32_t window, int normalize,
                   int ensure_pos, int has_table, uint32_t* d_out, uint32_t* d_out_count, double* d_L_out = nullptr);
int sort_dev_locked(pg_ctx* ctx, const double* d_scores, const uint32_t* d_seg, uint32_t n_seg, uint32_t n_items,
                    uint32_t max_seg, int desc, uint32_t* d_out);

void pipe_pool_destroy(pg_ctx* ctx);      // pipeline.hip
int scratch_reserve(pg_ctx* ctx, int slot, size_t bytes, void** out);
// raise a kernel's dynamic-LDS limit once per context (the attribute is per device: a process may hold
// contexts on several GPUs); caller holds ctx->mu
int ensure_dyn_lds(pg_ctx* ctx, const void* kernel, size_t bytes);
// features.hip: gather of integer feature columns, caller holds ctx->mu
int features_gather_i32_locked(pg_ctx* ctx, const pg_features* fs, const int32_t* col_idx, uint32_t n_cols,
                               const uint32_t* d_rows, uint32_t n, int32_t* d_out, const char* who);
}
