// common.hpp — shared host-side plumbing of libpairec_gpu.so (context, error convention).
// gfx950 (MI355X) only: wave = 64 lanes, 256 CUs, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pairec_gpu.h"

namespace pg {

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

}  // namespace pg

struct pg_table {
    float* d = nullptr;          // [rows][dim] fp32 row-major
    uint64_t rows = 0;
    uint32_t dim = 0;
    uint64_t row_offset = 0;     // global row id of local row 0 (sharded tables)
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
};

struct pg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    std::mutex mu;               // serialises calls on this context
    pg::Scratch scratch[10];     // named scratch slots (see users; 8 = pg_recommend_dnn3_dev's intermediates)
    std::mutex pipe_mu;          // serialises whole pg_recommend_* calls (they span several locked stages)
    std::map<const void*, size_t> dyn_lds;   // kernels whose dynamic-LDS limit was raised on this device
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_pool;   // per-launch timing events (scan kernel roofline figure)
    uint32_t last_scan_launches = 0;
    bool rank_timing_pending = false;
    pg_stats_t stats{};
    double last_scan_ms = 0.0;
    uint64_t last_scan_bytes = 0;
    // pinned host staging for small status words
    uint32_t* h_status = nullptr;
};

namespace pg {
int scratch_reserve(pg_ctx* ctx, int slot, size_t bytes, void** out);
// raise a kernel's dynamic-LDS limit once per context (the attribute is per device: a process may hold
// contexts on several GPUs); caller holds ctx->mu
int ensure_dyn_lds(pg_ctx* ctx, const void* kernel, size_t bytes);
// features.hip: gather of integer feature columns, caller holds ctx->mu
int features_gather_i32_locked(pg_ctx* ctx, const pg_features* fs, const int32_t* col_idx, uint32_t n_cols,
                               const uint32_t* d_rows, uint32_t n, int32_t* d_out, const char* who);
}
