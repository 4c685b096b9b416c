// api.cpp — context lifecycle, error convention, memory helpers of libpairec_gpu.so.
#include "common.hpp"

#include <cstdlib>

namespace pg {

static void knobs_from_env(Knobs* k) {
    auto num = [](const char* name, double dflt) { const char* v = getenv(name); return v ? atof(v) : dflt; };
    auto flag = [](const char* name) { return getenv(name) != nullptr; };
    k->screen_min = (uint32_t)num("PG_SCREEN_MIN", 0);
    k->recall_exact = flag("PG_RECALL_EXACT");
    k->pilot_fraction = num("PG_PILOT_FRACTION", 0.0);
    k->no_pilot = flag("PG_NO_PILOT");
    k->pilot_sigmas = num("PG_PILOT_SIGMAS", 6.0);
    k->chunk_growth = num("PG_CHUNK_GROWTH", 0.0);
    k->seed_rows = (uint32_t)num("PG_SEED_ROWS", 8192);
    k->pilot_growth = num("PG_PILOT_GROWTH", 0.0);
    k->debug_scan = flag("PG_DEBUG_SCAN");
    k->screen_bf16 = flag("PG_SCREEN_BF16");
    k->screen_i8 = flag("PG_SCREEN_I8");
    k->no_refine = flag("PG_NO_REFINE");
    k->refine_min_rows = (uint32_t)num("PG_REFINE_MIN_ROWS", (double)(1u << 24));
    k->no_screen_i4 = flag("PG_NO_SCREEN_I4");
    k->i4_max_lambda = num("PG_I4_MAX_LAMBDA", 1.7);
    k->i4_min_rows = (uint32_t)num("PG_I4_MIN_ROWS", (double)(1u << 22));
    k->no_screen_i4m = flag("PG_NO_SCREEN_I4M");
    k->i4m_max_queries = (uint32_t)num("PG_I4M_MAX_QUERIES", 64);
    k->i4m_min_queries = (uint32_t)num("PG_I4M_MIN_QUERIES", 1);
    k->i4m_max_lambda = num("PG_I4M_MAX_LAMBDA", 2.2);
    k->i4m_max_pairs = num("PG_I4M_MAX_PAIRS", 2.4e7);
    k->rank_no_ws = flag("PG_RANK_NO_WS");
    k->rank_sort_max = (uint32_t)num("PG_RANK_SORT_MAX", 32);
    k->rank_sort_work = num("PG_RANK_SORT_WORK", 7e7);
    k->split_sort_max = (uint32_t)num("PG_SPLIT_SORT_MAX", 96);
    k->sort_lds = flag("PG_SORT_LDS");
    k->fm2t_irs = flag("PG_FM2T_IRS");
    k->dpp_valu = flag("PG_DPP_VALU");
    k->max_rec_scale = (uint32_t)num("PG_MAX_REC_SCALE", 16);
    k->no_r2 = flag("PG_NO_R2");
    k->r2_min_factor = num("PG_R2_MIN_FACTOR", 3.0);
    k->coalescer_rejoin = !flag("PG_COALESCER_NO_REJOIN");
    k->coalescer_rejoin_us_per_caller = num("PG_COALESCER_REJOIN_US", 2.0);
    k->no_predict = flag("PG_NO_PREDICT");
    k->predict_sigmas = num("PG_PREDICT_SIGMAS", 4.5);
    k->predict_max_factor = num("PG_PREDICT_MAX_FACTOR", 4.0);
    k->predict_min_rows = (uint32_t)num("PG_PREDICT_MIN_ROWS", (double)(1u << 22));
    k->screen_early_share = (uint32_t)num("PG_SCREEN_EARLY_SHARE", 604);
    k->l2_exact = flag("PG_L2_EXACT");
    k->l2_max_slack = num("PG_L2_MAX_SLACK", 1.0);
    k->where_compact_max_rows = (uint32_t)num("PG_WHERE_COMPACT_MAX_ROWS", (double)(8u << 20));
    k->where_compact_min_ratio = (uint32_t)num("PG_WHERE_COMPACT_MIN_RATIO", 8);
    k->screen_early_share_narrow = (uint32_t)num("PG_SCREEN_EARLY_SHARE_NARROW", 512);
}

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int ensure_dyn_lds(pg_ctx* ctx, const void* kernel, size_t bytes) {
    auto it = ctx->dyn_lds.find(kernel);
    if (it != ctx->dyn_lds.end() && it->second >= bytes) return PG_OK;
    PG_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ctx->dyn_lds[kernel] = bytes;
    return PG_OK;
}

int scratch_reserve(pg_ctx* ctx, int slot, size_t bytes, void** out) {
    Scratch& s = ctx->scratch[slot];
    if (s.cap < bytes) {
        if (s.p) {
            PG_HIP(hipStreamSynchronize(ctx->stream));
            PG_HIP(hipFree(s.p));
            s.p = nullptr;
            s.cap = 0;
        }
        size_t cap = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
        PG_HIP(hipMalloc(&s.p, cap));
        s.cap = cap;
    }
    *out = s.p;
    return PG_OK;
}

}  // namespace pg

extern "C" {

const char* pg_last_error(void) { return pg::g_err.c_str(); }
const char* pg_version(void) { return "pairec_gpu 0.2 (gfx950)"; }

int pg_device_count(int* out) {
    PG_REQUIRE(out != nullptr, "pg_device_count: out is NULL");
    int n = 0;
    PG_HIP(hipGetDeviceCount(&n));
    *out = n;
    return PG_OK;
}

int pg_init(int device, void* stream, pg_ctx** out) {
    PG_REQUIRE(out != nullptr, "pg_init: out is NULL");
    int n = 0;
    PG_HIP(hipGetDeviceCount(&n));
    PG_REQUIRE(device >= 0 && device < n, "pg_init: device %d out of range (%d visible)", device, n);
    PG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pg::set_error("pg_init: device %d is %s; this library is built for gfx950 (MI355X) only",
                      device, prop.gcnArchName);
        return PG_ERR_UNSUPPORTED;
    }
    pg_ctx* c = new pg_ctx();
    pg::knobs_from_env(&c->knobs);
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        PG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    for (auto& e : c->ev) PG_HIP(hipEventCreate(&e));
    PG_HIP(hipHostMalloc((void**)&c->h_status, 4096));
    *out = c;
    return PG_OK;
}

int pg_shutdown(pg_ctx* ctx) {
    if (!ctx) return PG_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    pg::pipe_pool_destroy(ctx);
    for (auto& s : ctx->scratch)
        if (s.p) hipFree(s.p);
    for (auto& e : ctx->ev)
        if (e) hipEventDestroy(e);
    for (auto& e : ctx->ev_pool) hipEventDestroy(e);
    if (ctx->h_status) hipHostFree(ctx->h_status);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return PG_OK;
}

int pg_set_option(pg_ctx* ctx, const char* name, const char* value) {
    PG_REQUIRE(ctx && name && value, "pg_set_option: NULL argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::Knobs& k = ctx->knobs;
    const double v = atof(value);
    const bool b = v != 0.0;
    const std::string n(name);
    if (n == "screen_min") k.screen_min = (uint32_t)v;
    else if (n == "recall_exact") k.recall_exact = b;
    else if (n == "pilot_fraction") k.pilot_fraction = v;
    else if (n == "no_pilot") k.no_pilot = b;
    else if (n == "pilot_sigmas") k.pilot_sigmas = v;
    else if (n == "chunk_growth") k.chunk_growth = v;
    else if (n == "seed_rows") k.seed_rows = v >= 32 ? (uint32_t)v : 8192u;
    else if (n == "pilot_growth") k.pilot_growth = v;
    else if (n == "debug_scan") k.debug_scan = b;
    else if (n == "no_refine") k.no_refine = b;
    else if (n == "refine_min_rows") k.refine_min_rows = (uint32_t)v;
    else if (n == "no_screen_i4") k.no_screen_i4 = b;
    else if (n == "i4_max_lambda") k.i4_max_lambda = v;
    else if (n == "i4_min_rows") k.i4_min_rows = (uint32_t)v;
    else if (n == "no_screen_i4m") k.no_screen_i4m = b;
    else if (n == "i4m_max_queries") k.i4m_max_queries = (uint32_t)v;
    else if (n == "i4m_min_queries") k.i4m_min_queries = (uint32_t)v;
    else if (n == "i4m_max_lambda") k.i4m_max_lambda = v;
    else if (n == "i4m_max_pairs") k.i4m_max_pairs = v;
    else if (n == "rank_no_ws") k.rank_no_ws = b;
    else if (n == "rank_sort_max") k.rank_sort_max = (uint32_t)v;
    else if (n == "split_sort_max") k.split_sort_max = (uint32_t)v;
    else if (n == "rank_sort_work") k.rank_sort_work = v;
    else if (n == "sort_lds") k.sort_lds = b;
    else if (n == "stage_timers") ctx->timers_off = !b;      // (not a Knobs member: a coalescer's sibling context copies the knobs, its batches decide for themselves)
    else if (n == "fm2t_irs") k.fm2t_irs = b;
    else if (n == "dpp_valu") k.dpp_valu = b;
    else if (n == "no_r2") k.no_r2 = b;
    else if (n == "r2_min_factor") k.r2_min_factor = v;
    else if (n == "max_rec_scale") k.max_rec_scale = v >= 1 ? (uint32_t)v : 1u;
    else if (n == "coalescer_rejoin") k.coalescer_rejoin = b;
    else if (n == "coalescer_rejoin_us_per_caller") k.coalescer_rejoin_us_per_caller = v;
    else if (n == "no_predict") k.no_predict = b;
    else if (n == "predict_sigmas") k.predict_sigmas = v;
    else if (n == "predict_max_factor") k.predict_max_factor = v;
    else if (n == "predict_min_rows") k.predict_min_rows = (uint32_t)v;
    else if (n == "screen_early_share") k.screen_early_share = (uint32_t)v;
    else if (n == "l2_exact") k.l2_exact = b;
    else if (n == "l2_max_slack") k.l2_max_slack = v;
    else if (n == "where_compact_max_rows") k.where_compact_max_rows = (uint32_t)v;
    else if (n == "where_compact_min_ratio") k.where_compact_min_ratio = v >= 1 ? (uint32_t)v : 1u;
    else if (n == "screen_early_share_narrow") k.screen_early_share_narrow = (uint32_t)v;
    else {
        pg::set_error("pg_set_option: unknown option \"%s\"", name);
        return PG_ERR_INVALID;
    }
    return PG_OK;
}

int pg_synchronize(pg_ctx* ctx) {
    PG_REQUIRE(ctx, "pg_synchronize: ctx is NULL");
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_device_malloc(pg_ctx* ctx, size_t bytes, void** out) {
    PG_REQUIRE(ctx && out, "pg_device_malloc: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        pg::set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? PG_ERR_NOMEM : PG_ERR_DEVICE;
    }
    return PG_OK;
}

int pg_device_free(pg_ctx* ctx, void* p) {
    PG_REQUIRE(ctx, "pg_device_free: ctx is NULL");
    if (p) {
        PG_HIP(hipStreamSynchronize(ctx->stream));
        PG_HIP(hipFree(p));
    }
    return PG_OK;
}

int pg_memcpy_h2d(pg_ctx* ctx, void* dst, const void* src, size_t bytes) {
    PG_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pg_memcpy_h2d: NULL argument");
    if (bytes == 0) return PG_OK;
    PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_memcpy_d2h(pg_ctx* ctx, void* dst, const void* src, size_t bytes) {
    PG_REQUIRE(ctx && (bytes == 0 || (dst && src)), "pg_memcpy_d2h: NULL argument");
    if (bytes == 0) return PG_OK;
    PG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_stats(pg_ctx* ctx, pg_stats_t* out) {
    PG_REQUIRE(ctx && out, "pg_stats: NULL argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    if (ctx->rank_timing_pending) {                  // _dev rank calls only record their events
        PG_HIP(hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        PG_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
        ctx->stats.last_rank_ms = ms;
        ctx->rank_timing_pending = false;
    }
    *out = ctx->stats;
    return PG_OK;
}

int pg_last_scan_kernel_ms(pg_ctx* ctx, double* out_ms, uint64_t* out_bytes) {
    PG_REQUIRE(ctx, "pg_last_scan_kernel_ms: ctx is NULL");
    std::lock_guard<std::mutex> g(ctx->mu);
    if (out_ms) *out_ms = ctx->last_scan_ms;
    if (out_bytes) *out_bytes = ctx->last_scan_bytes;
    return PG_OK;
}

}  // extern "C"
