// group.hip — one process, several GPUs: the item table in contiguous row-range shards, one context per shard,
// the whole request batch behind ONE C call (BASELINE.json configs[4]).
//
// pairec itself is one Go process; a cgo host cannot join a torch.distributed job, so the sharded path is offered
// behind the C ABI as well (pairec_amd/dist.py stays the one-process-per-GPU harness bench.py uses under torchrun).
// The exchanges are KB..MB-sized and latency-bound, so there is no collective library in this path: with peer
// access enabled every shard stores straight into the buffers of the shard that needs the data (xGMI is fully
// connected — one hop), ordered by HIP events; on logical shards of one device the same stores are local.
//
//   every shard g      local exact top-k of its row range                                   (recall job, no host sync)
//   all-gather         shard g copies its [nq][k] (row, score) block into slot g of every shard's gather buffer
//   every shard h      identical deterministic merge → global top-k; compacts the candidates it OWNS, ranks them
//                      (DNN3, embedding rows are local), stores the scores into the lead's slab at their slots
//   lead (shard 0)     RankScore fusion → ItemRankScore sort → DPP candidates = first max(page, dpp_candidates) of
//                      the sorted list (DPPSort.doSort, sort/dpp_sort.go:280-291)
//   every shard h      stores the embedding rows it owns of those candidates into the lead's DPP buffer
//   lead               DPP greedy MAP (sort/dpp_sort.go:372-551) → the page; one device → host copy
// The host enqueues all of it without waiting and synchronises once, at the end; each shard's recall plan is
// verified then (a failed plan re-runs the step with that shard's fallback plan).
#include "pipeline.hpp"

#include <algorithm>

namespace pg {
namespace {

struct Shard {
    pg_ctx* ctx = nullptr;
    pg_table* tab = nullptr;
    pg_model* model = nullptr;
    PipeRun* run = nullptr;
    bool plan_failed = false;        // the last step's recall plan did not hold: the retry runs this shard's next plan
    hipEvent_t ev_lists = nullptr, ev_rank = nullptr, ev_emb = nullptr;
    // sized for (nq_cap, k_cap)
    float* d_q = nullptr;
    uint64_t* d_rows = nullptr;      // local top-k [nq][k]
    float* d_sc = nullptr;
    uint64_t* g_rows = nullptr;      // gathered [G][nq][k]
    float* g_sc = nullptr;
    uint64_t* m_rows = nullptr;      // merged [nq][k]
    float* m_sc = nullptr;
    uint32_t* d_local = nullptr;     // compacted local rows of the candidates this shard owns
    uint32_t* d_slot = nullptr;      // their positions q * k + j in the merged lists
    uint32_t* d_off = nullptr;       // [nq + 1]
    float* d_rank = nullptr;         // compacted model scores
};

__global__ void owned_count_kernel(const uint64_t* __restrict__ rows, uint32_t k, uint64_t off, uint64_t nrows,
                                   uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s[256];
    const uint32_t q = blockIdx.x;
    uint32_t c = 0;
    for (uint32_t j = threadIdx.x; j < k; j += 256) {
        const uint64_t r = rows[(size_t)q * k + j];
        c += (r != ~0ull && r >= off && r - off < nrows) ? 1u : 0u;
    }
    s[threadIdx.x] = c;
    __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[q] = s[0];
}

// off[0..nq] = exclusive scan of cnt[0..nq)  (nq <= 256)
__global__ void owned_scan_kernel(const uint32_t* __restrict__ cnt, uint32_t nq, uint32_t* __restrict__ off) {
    __shared__ uint32_t s[256];
    const uint32_t t = threadIdx.x;
    const uint32_t v = t < nq ? cnt[t] : 0u;
    s[t] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t x = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += x;
        __syncthreads();
    }
    if (t < nq) off[t] = s[t] - v;
    if (t == nq - 1) off[nq] = s[t];
}

// stable compaction of request q's owned candidates: local row and slot, at off[q] + rank among the owned
__global__ __launch_bounds__(1024) void owned_fill_kernel(const uint64_t* __restrict__ rows, uint32_t k, uint64_t off,
                                                          uint64_t nrows, const uint32_t* __restrict__ req_off,
                                                          uint32_t* __restrict__ local, uint32_t* __restrict__ slot) {
    __shared__ uint32_t s[1024];
    __shared__ uint32_t base;
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    if (t == 0) base = req_off[q];
    __syncthreads();
    for (uint32_t j0 = 0; j0 < k; j0 += 1024) {
        const uint32_t j = j0 + t;
        uint64_t r = ~0ull;
        if (j < k) r = rows[(size_t)q * k + j];
        const uint32_t mine = (r != ~0ull && r >= off && r - off < nrows) ? 1u : 0u;
        s[t] = mine;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const uint32_t x = t >= d ? s[t - d] : 0u;
            __syncthreads();
            s[t] += x;
            __syncthreads();
        }
        if (mine) {
            const uint32_t p = base + s[t] - 1;
            local[p] = (uint32_t)(r - off);
            slot[p] = q * k + j;
        }
        __syncthreads();
        if (t == 0) base += s[1023];
        __syncthreads();
    }
}

// slab[slot[i]] = score[i] for the owned candidates (slab may live on the lead's device: peer store)
__global__ void scatter_scores_kernel(const float* __restrict__ score, const uint32_t* __restrict__ slot,
                                      const uint32_t* __restrict__ total, uint32_t cap, float* __restrict__ slab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap && i < *total) slab[slot[i]] = score[i];
}

}  // namespace
}  // namespace pg

struct pg_group {
    std::vector<pg::Shard> sh;
    std::vector<int> dev;
    uint64_t total_rows = 0;
    uint32_t dim = 0;
    uint32_t nq_cap = 0, k_cap = 0, c_cap = 0, top_cap = 0, nv_cap = 0;
    std::mutex mu;                 // one step at a time
    hipEvent_t ev_ready = nullptr, ev_sel = nullptr, ev_done = nullptr;
    // lead-only buffers
    float* slab = nullptr;
    double* vars = nullptr;
    double* fused = nullptr;
    uint32_t* order = nullptr;
    uint32_t* seg = nullptr;
    uint32_t* d_err = nullptr;
    uint32_t* d_count = nullptr;
    uint64_t* c_rows = nullptr;
    double* c_rel = nullptr;
    float* c_emb = nullptr;
    uint32_t* pick = nullptr;
    uint32_t* pick_cnt = nullptr;
    char* d_page = nullptr;
    char* h_page = nullptr;        // pinned
    uint32_t* h_flags = nullptr;   // pinned: RankScore flags per request + counts
};

namespace pg {
namespace {

void shard_range(uint64_t total, uint32_t G, uint32_t g, uint64_t* b, uint64_t* e) {
    const uint64_t base = total / G, rem = total % G;
    *b = g * base + std::min<uint64_t>(g, rem);
    *e = *b + base + (g < rem ? 1 : 0);
}

void free_step_buffers(pg_group* g) {
    for (auto& s : g->sh) {
        hipSetDevice(s.ctx->device);
        for (void* p : {(void*)s.d_q, (void*)s.d_rows, (void*)s.d_sc, (void*)s.g_rows, (void*)s.g_sc, (void*)s.m_rows, (void*)s.m_sc,
                        (void*)s.d_local, (void*)s.d_slot, (void*)s.d_off, (void*)s.d_rank})
            if (p) hipFree(p);
        s.d_q = nullptr; s.d_rows = nullptr; s.d_sc = nullptr; s.g_rows = nullptr; s.g_sc = nullptr; s.m_rows = nullptr;
        s.m_sc = nullptr; s.d_local = nullptr; s.d_slot = nullptr; s.d_off = nullptr; s.d_rank = nullptr;
    }
    if (g->sh.empty()) return;
    hipSetDevice(g->sh[0].ctx->device);
    for (void* p : {(void*)g->slab, (void*)g->vars, (void*)g->fused, (void*)g->order, (void*)g->seg, (void*)g->d_err, (void*)g->d_count,
                    (void*)g->c_rows, (void*)g->c_rel, (void*)g->c_emb, (void*)g->pick, (void*)g->pick_cnt, (void*)g->d_page})
        if (p) hipFree(p);
    if (g->h_page) hipHostFree(g->h_page);
    if (g->h_flags) hipHostFree(g->h_flags);
    g->slab = nullptr; g->vars = nullptr; g->fused = nullptr; g->order = nullptr; g->seg = nullptr; g->d_err = nullptr;
    g->d_count = nullptr; g->c_rows = nullptr; g->c_rel = nullptr; g->c_emb = nullptr; g->pick = nullptr; g->pick_cnt = nullptr;
    g->d_page = nullptr; g->h_page = nullptr; g->h_flags = nullptr;
    g->nq_cap = g->k_cap = g->c_cap = g->top_cap = g->nv_cap = 0;
}

int ensure_step_buffers(pg_group* g, uint32_t nq, uint32_t k, uint32_t C, uint32_t top_n, int nv) {
    if (nq <= g->nq_cap && k <= g->k_cap && C <= g->c_cap && top_n <= g->top_cap && (uint32_t)nv <= g->nv_cap) return PG_OK;
    for (auto& s : g->sh) {
        hipSetDevice(s.ctx->device);
        hipStreamSynchronize(s.ctx->stream);
    }
    // grow every dimension to the largest seen so far: alternating shapes must not re-allocate on every call, and a
    // later expression with more variables must not find `vars` sized for the first one's
    nq = std::max(nq, g->nq_cap); k = std::max(k, g->k_cap); C = std::max(C, g->c_cap); top_n = std::max(top_n, g->top_cap);
    nv = std::max(nv, (int)g->nv_cap);
    free_step_buffers(g);
    const uint32_t G = (uint32_t)g->sh.size();
    const size_t n = (size_t)nq * k;
    for (auto& s : g->sh) {
        PG_HIP(hipSetDevice(s.ctx->device));
        PG_HIP(hipMalloc((void**)&s.d_q, (size_t)nq * g->dim * 4));
        PG_HIP(hipMalloc((void**)&s.d_rows, n * 8));
        PG_HIP(hipMalloc((void**)&s.d_sc, n * 4));
        PG_HIP(hipMalloc((void**)&s.g_rows, n * 8 * G));
        PG_HIP(hipMalloc((void**)&s.g_sc, n * 4 * G));
        PG_HIP(hipMalloc((void**)&s.m_rows, n * 8));
        PG_HIP(hipMalloc((void**)&s.m_sc, n * 4));
        PG_HIP(hipMalloc((void**)&s.d_local, n * 4));
        PG_HIP(hipMalloc((void**)&s.d_slot, n * 4));
        PG_HIP(hipMalloc((void**)&s.d_off, ((size_t)nq + 1 + 256) * 4));
        PG_HIP(hipMalloc((void**)&s.d_rank, n * 4));
    }
    PG_HIP(hipSetDevice(g->sh[0].ctx->device));
    const size_t page = (size_t)nq * top_n * 24;
    PG_HIP(hipMalloc((void**)&g->slab, n * 4));
    PG_HIP(hipMalloc((void**)&g->vars, n * 8 * (size_t)std::max(nv, 1)));
    PG_HIP(hipMalloc((void**)&g->fused, n * 8));
    PG_HIP(hipMalloc((void**)&g->order, n * 4));
    PG_HIP(hipMalloc((void**)&g->seg, ((size_t)nq + 1) * 4));
    PG_HIP(hipMalloc((void**)&g->d_err, 256 * 4));
    PG_HIP(hipMalloc((void**)&g->d_count, 256 * 4));
    PG_HIP(hipMalloc((void**)&g->c_rows, (size_t)nq * std::max(C, 1u) * 8));
    PG_HIP(hipMalloc((void**)&g->c_rel, (size_t)nq * std::max(C, 1u) * 8));
    PG_HIP(hipMalloc((void**)&g->c_emb, (size_t)nq * std::max(C, 1u) * g->dim * 4));
    PG_HIP(hipMalloc((void**)&g->pick, (size_t)nq * top_n * 4));
    PG_HIP(hipMalloc((void**)&g->pick_cnt, 256 * 4));
    PG_HIP(hipMalloc((void**)&g->d_page, page));
    PG_HIP(hipHostMalloc((void**)&g->h_page, page));
    PG_HIP(hipHostMalloc((void**)&g->h_flags, 1024 * 4));
    g->nq_cap = nq; g->k_cap = k; g->c_cap = C; g->top_cap = top_n; g->nv_cap = (uint32_t)std::max(nv, 1);
    return PG_OK;
}

// var_src bit i = 1: variable i is the model's score
__global__ void group_bind_vars_kernel(const float* __restrict__ recall, const float* __restrict__ rank, uint32_t n,
                                       uint32_t nv, uint32_t src_mask, double* __restrict__ vars) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = (double)recall[i], b = (double)rank[i];
    for (uint32_t v = 0; v < nv; ++v) vars[(size_t)v * n + i] = ((src_mask >> v) & 1u) ? b : a;
}

}  // namespace
}  // namespace pg

extern "C" {

// ---- the shard-side steps as device-level calls (pairec_amd/dist.py drives them around torch.distributed collectives) ----
int pg_owned_compact_dev(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t nq, uint32_t k,
                         uint32_t* d_local, uint32_t* d_slot, uint32_t* d_req_offsets) {
    PG_REQUIRE(ctx && t && d_rows && d_local && d_slot && d_req_offsets, "pg_owned_compact_dev: NULL argument");
    PG_REQUIRE(nq >= 1 && nq <= (uint32_t)pg::kMaxQueries && k >= 1, "pg_owned_compact_dev: bad nq / k");
    std::lock_guard<std::mutex> g(ctx->mu);
    void* p;
    int rc;
    if ((rc = pg::scratch_reserve(ctx, 9, 4096, &p))) return rc;
    uint32_t* d_cnt = (uint32_t*)p;
    pg::owned_count_kernel<<<nq, 256, 0, ctx->stream>>>(d_rows, k, t->row_offset, t->rows, d_cnt);
    pg::owned_scan_kernel<<<1, 256, 0, ctx->stream>>>(d_cnt, nq, d_req_offsets);
    pg::owned_fill_kernel<<<nq, 1024, 0, ctx->stream>>>(d_rows, k, t->row_offset, t->rows, d_req_offsets, d_local, d_slot);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

int pg_scatter_f32_dev(pg_ctx* ctx, const float* d_vals, const uint32_t* d_slot, const uint32_t* d_total, uint32_t cap,
                       float* d_out) {
    PG_REQUIRE(ctx && d_vals && d_slot && d_total && d_out, "pg_scatter_f32_dev: NULL argument");
    if (cap == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::scatter_scores_kernel<<<(cap + 255) / 256, 256, 0, ctx->stream>>>(d_vals, d_slot, d_total, cap, d_out);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

int pg_dpp_candidates_dev(pg_ctx* ctx, const uint32_t* d_order, const uint64_t* d_rows, const double* d_fused, uint32_t nq,
                          uint32_t k, uint32_t n_cand, uint64_t* d_c_rows, double* d_c_rel) {
    PG_REQUIRE(ctx && d_order && d_rows && d_fused && d_c_rows && d_c_rel, "pg_dpp_candidates_dev: NULL argument");
    PG_REQUIRE(n_cand >= 1 && n_cand <= k, "pg_dpp_candidates_dev: n_cand %u outside 1..k", n_cand);
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::sorted_head_launch(ctx->stream, d_order, d_rows, d_fused, nq, k, n_cand, d_c_rows, d_c_rel);
}

int pg_gather_owned_rows_dev(pg_ctx* ctx, const pg_table* t, const uint64_t* d_global_rows, uint32_t n, float* d_out) {
    PG_REQUIRE(ctx && t && (n == 0 || (d_global_rows && d_out)), "pg_gather_owned_rows_dev: NULL argument");
    if (n == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::gather_global_rows_launch(ctx->stream, t, d_global_rows, n, d_out);
}

int pg_dpp_batch_dev(pg_ctx* ctx, const float* d_emb, const double* d_rel, uint32_t n_req, uint32_t n, uint32_t dim,
                     double alpha, uint32_t topn, uint32_t window, int normalize_emb, uint32_t* d_out_idx,
                     uint32_t* d_out_count) {
    PG_REQUIRE(ctx && d_emb && d_rel && d_out_idx && d_out_count, "pg_dpp_batch_dev: NULL argument");
    PG_REQUIRE(n <= 8192 && dim <= 4096 && n_req <= 65535, "pg_dpp_batch_dev: %u requests x %u candidates x %u dims unsupported", n_req, n, dim);
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::dpp_run_locked(ctx, d_emb, nullptr, d_rel, n_req, n, dim, 0, alpha, topn, window, normalize_emb, 1, 1, d_out_idx, d_out_count);
}

int pg_topk_merge_lists_dev(pg_ctx* ctx, const uint64_t* d_rows, const float* d_scores, uint32_t nq, uint32_t nlists,
                            uint32_t per_list, int list_major, uint32_t k, uint64_t* d_out_rows, float* d_out_scores) {
    PG_REQUIRE(ctx && d_rows && d_scores && d_out_rows && d_out_scores, "pg_topk_merge_lists_dev: NULL argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::topk_merge_locked(ctx, d_rows, d_scores, nq, nlists, per_list, list_major, k, d_out_rows, d_out_scores, nullptr);
}

int pg_group_create(const int* devices, uint32_t n_shards, pg_group** out) {
    PG_REQUIRE(devices && out && n_shards >= 1 && n_shards <= 64, "pg_group_create: bad argument");
    pg_group* g = new pg_group();
    g->sh.resize(n_shards);
    g->dev.assign(devices, devices + n_shards);
    int rc = PG_OK;
    for (uint32_t i = 0; i < n_shards && !rc; ++i) {
        rc = pg_init(devices[i], nullptr, &g->sh[i].ctx);
        if (rc) break;
        auto& s = g->sh[i];
        if (hipEventCreateWithFlags(&s.ev_lists, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_rank, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_emb, hipEventDisableTiming) != hipSuccess) {
            pg::set_error("pg_group_create: %s", hipGetErrorString(hipGetLastError()));
            rc = PG_ERR_DEVICE;
            break;
        }
        rc = pg::pipe_run_acquire(s.ctx, &s.run);
    }
    // peer access between every pair of distinct devices: the exchanges are direct stores / copies over xGMI
    for (uint32_t i = 0; i < n_shards && !rc; ++i)
        for (uint32_t j = 0; j < n_shards && !rc; ++j) {
            if (devices[i] == devices[j]) continue;
            int can = 0;
            hipDeviceCanAccessPeer(&can, devices[i], devices[j]);
            if (!can) {
                pg::set_error("pg_group_create: device %d cannot access device %d (no peer path)", devices[i], devices[j]);
                rc = PG_ERR_UNSUPPORTED;
                break;
            }
            hipSetDevice(devices[i]);
            const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                pg::set_error("pg_group_create: hipDeviceEnablePeerAccess(%d -> %d): %s", devices[i], devices[j], hipGetErrorString(e));
                rc = PG_ERR_DEVICE;
            }
            (void)hipGetLastError();
        }
    if (!rc) {
        hipSetDevice(devices[0]);
        if (hipEventCreateWithFlags(&g->ev_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_sel, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming) != hipSuccess) {
            pg::set_error("pg_group_create: %s", hipGetErrorString(hipGetLastError()));
            rc = PG_ERR_DEVICE;
        }
    }
    if (rc) {
        pg_group_destroy(g);
        return rc;
    }
    *out = g;
    return PG_OK;
}

int pg_group_destroy(pg_group* g) {
    if (!g) return PG_OK;
    for (auto& s : g->sh)
        if (s.ctx) {
            hipSetDevice(s.ctx->device);
            hipStreamSynchronize(s.ctx->stream);
        }
    pg::free_step_buffers(g);
    for (hipEvent_t e : {g->ev_ready, g->ev_sel, g->ev_done})
        if (e) hipEventDestroy(e);
    for (auto& s : g->sh) {
        if (!s.ctx) continue;
        hipSetDevice(s.ctx->device);
        for (hipEvent_t e : {s.ev_lists, s.ev_rank, s.ev_emb})
            if (e) hipEventDestroy(e);
        if (s.run) pg::pipe_run_release(s.ctx, s.run);
        if (s.model) pg_model_destroy(s.ctx, s.model);
        if (s.tab) pg_table_destroy(s.ctx, s.tab);
        pg_shutdown(s.ctx);
    }
    delete g;
    return PG_OK;
}

uint32_t pg_group_size(const pg_group* g) { return g ? (uint32_t)g->sh.size() : 0; }
pg_ctx* pg_group_ctx(pg_group* g, uint32_t shard) { return g && shard < g->sh.size() ? g->sh[shard].ctx : nullptr; }
pg_table* pg_group_table(pg_group* g, uint32_t shard) { return g && shard < g->sh.size() ? g->sh[shard].tab : nullptr; }

int pg_group_table_create(pg_group* g, uint64_t total_rows, uint32_t dim) {
    PG_REQUIRE(g && total_rows >= g->sh.size(), "pg_group_table_create: bad argument");
    PG_REQUIRE(total_rows < 0xFFFFFFFFull, "pg_group_table_create: global row ids must stay below 2^32");
    std::lock_guard<std::mutex> lk(g->mu);
    const uint32_t G = (uint32_t)g->sh.size();
    for (uint32_t i = 0; i < G; ++i) {
        auto& s = g->sh[i];
        if (s.tab) {
            pg_table_destroy(s.ctx, s.tab);
            s.tab = nullptr;
        }
        uint64_t b, e;
        pg::shard_range(total_rows, G, i, &b, &e);
        int rc;
        if ((rc = pg_table_create(s.ctx, e - b, dim, b, &s.tab))) return rc;
    }
    g->total_rows = total_rows;
    g->dim = dim;
    return PG_OK;
}

int pg_group_table_fill_synthetic(pg_group* g, uint64_t seed, int normalize) {
    PG_REQUIRE(g && g->total_rows, "pg_group_table_fill_synthetic: no table");
    std::lock_guard<std::mutex> lk(g->mu);
    for (auto& s : g->sh) {
        int rc;
        if ((rc = pg_table_fill_synthetic(s.ctx, s.tab, seed, normalize))) return rc;
    }
    return PG_OK;
}

int pg_group_table_upload(pg_group* g, uint64_t row0, uint64_t nrows, const float* host_rows) {
    PG_REQUIRE(g && g->total_rows && (nrows == 0 || host_rows), "pg_group_table_upload: bad argument");
    PG_REQUIRE(row0 + nrows <= g->total_rows, "pg_group_table_upload: rows %llu..%llu outside the table",
               (unsigned long long)row0, (unsigned long long)(row0 + nrows));
    std::lock_guard<std::mutex> lk(g->mu);
    const uint32_t G = (uint32_t)g->sh.size();
    for (uint32_t i = 0; i < G; ++i) {
        uint64_t b, e;
        pg::shard_range(g->total_rows, G, i, &b, &e);
        const uint64_t lo = std::max(b, row0), hi = std::min(e, row0 + nrows);
        if (lo >= hi) continue;
        int rc;
        if ((rc = pg_table_upload(g->sh[i].ctx, g->sh[i].tab, lo - b, hi - lo, host_rows + (lo - row0) * g->dim))) return rc;
    }
    return PG_OK;
}

int pg_group_model_load(pg_group* g, pg_model_kind kind, pg_prec prec, const void* blob, size_t len) {
    PG_REQUIRE(g && blob, "pg_group_model_load: NULL argument");
    std::lock_guard<std::mutex> lk(g->mu);
    for (auto& s : g->sh) {                      // weights are replicated (0.5 MB)
        if (s.model) {
            pg_model_destroy(s.ctx, s.model);
            s.model = nullptr;
        }
        int rc;
        if ((rc = pg_model_load(s.ctx, kind, prec, blob, len, &s.model))) return rc;
    }
    return PG_OK;
}

int pg_group_recommend(pg_group* g, const pg_expr* e, const char* rank_var, const pg_group_plan* plan,
                       const float* user_vecs, uint32_t nq, uint32_t top_n, uint64_t* out_rows,
                       float* out_recall_scores, float* out_rank_scores, double* out_fused, uint32_t* out_count) {
    PG_REQUIRE(g && e && rank_var && plan && user_vecs && out_rows && out_recall_scores && out_rank_scores && out_fused,
               "pg_group_recommend: NULL argument");
    PG_REQUIRE(g->total_rows && g->sh[0].model, "pg_group_recommend: the group has no table or no model");
    const uint32_t k = plan->k, G = (uint32_t)g->sh.size();
    PG_REQUIRE(nq >= 1 && nq <= (uint32_t)pg::kMaxQueries && k >= 1 && k <= 16384, "pg_group_recommend: bad nq / k");
    PG_REQUIRE(top_n >= 1 && top_n <= k, "pg_group_recommend: top_n %u outside 1..k", top_n);
    PG_REQUIRE(g->total_rows >= k, "pg_group_recommend: the table has fewer rows than k");
    PG_REQUIRE((uint64_t)G * k <= (uint64_t)k + (1u << 19), "pg_group_recommend: %u shards x k=%u exceeds the merge capacity", G, k);
    const pg_model* m0 = g->sh[0].model;
    PG_REQUIRE(m0->kind == PG_MODEL_DNN3 && m0->d_item == g->dim && m0->d_user == g->dim,
               "pg_group_recommend: the model must be DNN3 with d_user = d_item = the table's dim");
    const uint32_t C = plan->dpp_candidates ? std::min(k, std::max(top_n, plan->dpp_candidates)) : 0u;
    PG_REQUIRE(C <= 8192, "pg_group_recommend: %u DPP candidates (at most 8192)", C);
    std::vector<int> src;
    int rc;
    if ((rc = pg::recommend_bind_vars(e, &rank_var, 1, &src, "pg_group_recommend"))) return rc;
    const int nv = (int)src.size();
    uint32_t mask = 0;
    for (int i = 0; i < nv; ++i) mask |= (src[(size_t)i] >= 0 ? 1u : 0u) << i;

    std::lock_guard<std::mutex> lk(g->mu);
    if ((rc = pg::ensure_step_buffers(g, nq, k, C, top_n, nv))) return rc;
    const uint32_t n = nq * k;
    pg::Shard& lead = g->sh[0];
    hipStream_t ls = lead.ctx->stream;

    for (uint32_t attempt = 0;; ++attempt) {
        // ---- 1. local recall on every shard -----------------------------------------------------------------
        for (uint32_t i = 0; i < G; ++i) {
            pg::Shard& s = g->sh[i];
            PG_HIP(hipSetDevice(s.ctx->device));
            PG_HIP(hipMemcpyAsync(s.d_q, user_vecs, (size_t)nq * g->dim * 4, hipMemcpyHostToDevice, s.ctx->stream));
            std::lock_guard<std::mutex> cg(s.ctx->mu);
            pg::RecallJob& j = s.run->job;
            const bool retry_next_plan = attempt > 0 && s.plan_failed;
            if (!retry_next_plan) {
                j = pg::RecallJob();
                j.ctx = s.ctx;
                j.t = s.tab;
                j.d_queries = s.d_q;
                j.nq = nq;
                j.k = k;
                j.d_out_rows = s.d_rows;
                j.d_out_scores = s.d_sc;
                j.h_status = s.run->h_status;
                j.events = &s.run->events;
                if ((rc = pg::recall_job_prepare(&j))) return rc;
            }
            if ((rc = pg::recall_job_enqueue(&j))) return rc;
        }
        // ---- 2. all-gather of the per-shard lists: shard i stores its block into slot i of every shard ----------
        for (uint32_t i = 0; i < G; ++i) {
            pg::Shard& s = g->sh[i];
            PG_HIP(hipSetDevice(s.ctx->device));
            for (uint32_t h = 0; h < G; ++h) {
                PG_HIP(hipMemcpyAsync(g->sh[h].g_rows + (size_t)i * n, s.d_rows, (size_t)n * 8, hipMemcpyDeviceToDevice, s.ctx->stream));
                PG_HIP(hipMemcpyAsync(g->sh[h].g_sc + (size_t)i * n, s.d_sc, (size_t)n * 4, hipMemcpyDeviceToDevice, s.ctx->stream));
            }
            PG_HIP(hipEventRecord(s.ev_lists, s.ctx->stream));
        }
        // the lead's slab starts from zero (padding slots have no owner)
        PG_HIP(hipSetDevice(lead.ctx->device));
        PG_HIP(hipMemsetAsync(g->slab, 0, (size_t)n * 4, ls));
        PG_HIP(hipEventRecord(g->ev_ready, ls));
        // ---- 3. every shard: merge, rank what it owns, store the scores into the lead's slab -------------------
        for (uint32_t h = 0; h < G; ++h) {
            pg::Shard& s = g->sh[h];
            PG_HIP(hipSetDevice(s.ctx->device));
            hipStream_t st = s.ctx->stream;
            for (uint32_t i = 0; i < G; ++i)
                if (i != h) PG_HIP(hipStreamWaitEvent(st, g->sh[i].ev_lists, 0));
            {
                std::lock_guard<std::mutex> cg(s.ctx->mu);
                if ((rc = pg::topk_merge_locked(s.ctx, s.g_rows, s.g_sc, nq, G, k, 1, k, s.m_rows, s.m_sc, h == 0 ? g->d_count : nullptr))) return rc;
                uint32_t* d_cnt = s.d_off + nq + 1;
                pg::owned_count_kernel<<<nq, 256, 0, st>>>(s.m_rows, k, s.tab->row_offset, s.tab->rows, d_cnt);
                pg::owned_scan_kernel<<<1, 256, 0, st>>>(d_cnt, nq, s.d_off);
                pg::owned_fill_kernel<<<nq, 1024, 0, st>>>(s.m_rows, k, s.tab->row_offset, s.tab->rows, s.d_off, s.d_local, s.d_slot);
                PG_HIP(hipGetLastError());
                if ((rc = pg::rank_dnn3_dev_locked(s.ctx, s.model, s.tab, s.d_q, s.d_local, s.d_off, nq, n, s.d_rank))) return rc;
            }
            PG_HIP(hipStreamWaitEvent(st, g->ev_ready, 0));
            pg::scatter_scores_kernel<<<(n + 255) / 256, 256, 0, st>>>(s.d_rank, s.d_slot, s.d_off + nq, n, g->slab);
            PG_HIP(hipGetLastError());
            PG_HIP(hipEventRecord(s.ev_rank, st));
        }
        // ---- 4. lead: fusion, sort, DPP candidates ------------------------------------------------------------------
        PG_HIP(hipSetDevice(lead.ctx->device));
        for (uint32_t h = 1; h < G; ++h) PG_HIP(hipStreamWaitEvent(ls, g->sh[h].ev_rank, 0));
        {
            std::lock_guard<std::mutex> cg(lead.ctx->mu);
            if ((rc = pg::uniform_offsets_locked(lead.ctx, nq, k, g->seg))) return rc;
            if (nv > 0) {
                pg::group_bind_vars_kernel<<<(n + 255) / 256, 256, 0, ls>>>(lead.m_sc, g->slab, n, (uint32_t)nv, mask, g->vars);
                PG_HIP(hipGetLastError());
            }
            PG_HIP(hipMemsetAsync(g->d_err, 0, 256 * 4, ls));
            if ((rc = pg::expr_eval_enqueue_locked(lead.ctx, e, g->vars, n, g->fused, g->d_err, k))) return rc;
            if ((rc = pg::sort_dev_locked(lead.ctx, g->fused, g->seg, nq, n, k, 1, g->order))) return rc;
            if (C) {
                if ((rc = pg::sorted_head_launch(ls, g->order, lead.m_rows, g->fused, nq, k, C, g->c_rows, g->c_rel))) return rc;
            }
        }
        if (C) {
            PG_HIP(hipEventRecord(g->ev_sel, ls));
            // ---- 5. every shard stores the candidate embeddings it owns into the lead's DPP buffer ------------------
            for (uint32_t h = 0; h < G; ++h) {
                pg::Shard& s = g->sh[h];
                PG_HIP(hipSetDevice(s.ctx->device));
                hipStream_t st = s.ctx->stream;
                if (h) PG_HIP(hipStreamWaitEvent(st, g->ev_sel, 0));
                if ((rc = pg::gather_global_rows_launch(st, s.tab, g->c_rows, nq * C, g->c_emb))) return rc;
                if (h) PG_HIP(hipEventRecord(s.ev_emb, st));
            }
            PG_HIP(hipSetDevice(lead.ctx->device));
            for (uint32_t h = 1; h < G; ++h) PG_HIP(hipStreamWaitEvent(ls, g->sh[h].ev_emb, 0));
            std::lock_guard<std::mutex> cg(lead.ctx->mu);
            if ((rc = pg::dpp_run_locked(lead.ctx, g->c_emb, nullptr, g->c_rel, nq, C, g->dim, 0, plan->dpp_alpha, top_n,
                                         plan->dpp_window, plan->dpp_normalize_emb, 1, 1, g->pick, g->pick_cnt)))
                return rc;
        }
        // ---- 6. the page ------------------------------------------------------------------------------------------------
        const size_t np = (size_t)nq * top_n;
        if ((rc = pg::page_launch(ls, g->order, C ? g->pick : nullptr, g->pick_cnt, lead.m_rows, lead.m_sc, g->slab, (size_t)n, 1, g->fused, nq, k,
                                  top_n, g->d_page)))
            return rc;
        PG_HIP(hipMemcpyAsync(g->h_page, g->d_page, np * 24, hipMemcpyDeviceToHost, ls));
        PG_HIP(hipMemcpyAsync(g->h_flags, g->d_err, (size_t)nq * 4, hipMemcpyDeviceToHost, ls));
        PG_HIP(hipMemcpyAsync(g->h_flags + 256, g->d_count, (size_t)nq * 4, hipMemcpyDeviceToHost, ls));
        if (C) PG_HIP(hipMemcpyAsync(g->h_flags + 512, g->pick_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, ls));
        PG_HIP(hipEventRecord(g->ev_done, ls));
        // ---- the one host synchronisation of the step, then the deferred verification of every shard's plan -------------------
        PG_HIP(hipEventSynchronize(g->ev_done));
        bool all_ok = true;
        for (uint32_t i = 0; i < G; ++i) {
            pg::Shard& s = g->sh[i];
            PG_HIP(hipSetDevice(s.ctx->device));
            PG_HIP(hipStreamSynchronize(s.ctx->stream));
            std::lock_guard<std::mutex> cg(s.ctx->mu);
            bool ok = false;
            if ((rc = pg::recall_job_check(&s.run->job, &ok))) return rc;
            s.plan_failed = !ok;                              // remembered for the retry: this shard moves to its next plan
            if (ok) pg::recall_job_finish(&s.run->job);
            all_ok = all_ok && ok;
        }
        if (all_ok) break;
        if (attempt >= 3) {
            pg::set_error("pg_group_recommend: recall plans kept failing (internal error)");
            return PG_ERR_DEVICE;
        }
    }
    for (uint32_t q = 0; q < nq; ++q)
        if (g->h_flags[q]) {
            pg::set_expr_arith_error(e);
            return PG_ERR_ARITH;
        }
    const size_t np = (size_t)nq * top_n;
    const uint64_t* p_rows = (const uint64_t*)g->h_page;
    const double* p_fused = (const double*)(p_rows + np);
    const float* p_recall = (const float*)(p_fused + np);
    const float* p_rank = p_recall + np;
    memcpy(out_rows, p_rows, np * 8);
    memcpy(out_fused, p_fused, np * 8);
    memcpy(out_recall_scores, p_recall, np * 4);
    memcpy(out_rank_scores, p_rank, np * 4);
    if (out_count)
        for (uint32_t q = 0; q < nq; ++q) out_count[q] = C ? g->h_flags[512 + q] : std::min(top_n, g->h_flags[256 + q]);
    return PG_OK;
}

}  // extern "C"
