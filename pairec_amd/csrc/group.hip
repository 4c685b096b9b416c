// group.hip — one process, several GPUs: the item table in contiguous row-range shards, the whole request batch behind
// ONE C call (BASELINE.json configs[4]).
//
// pairec itself is one Go process; a cgo host cannot join a torch.distributed job, so the sharded path is offered
// behind the C ABI as well (pairec_amd/dist.py stays the one-process-per-GPU harness bench.py uses under torchrun).
// The exchanges are KB..MB-sized and latency-bound, so there is no collective library in this path: with peer
// access enabled every shard stores straight into the buffers of the shard that needs the data (xGMI is fully
// connected — one hop), ordered by HIP events; on logical shards of one device the same stores are local.
//
//   every shard g      local exact top-k of its row range                                   (recall job, no host sync)
//   all-gather         shard g copies the HEADS of its lists — the best m = ceil(k/G + 6 sqrt(k/G) + 8) entries of every request,
//                      packed (rows | scores), ONE copy per peer — into slot g of every shard's gather buffer (G = 8, k = 5 000:
//                      783 entries, 2.4 MB per shard and 256-request step instead of 15.4).  Exact: an entry a shard did not send
//                      ranks behind that shard's m-th, so nothing unsent can belong to the merged top-k unless some shard's m-th
//                      entry lies strictly inside it — which every shard checks after its merge (tail_needed_kernel); the flag
//                      travels with the step's status words and _end repeats the step with the full lists (counted:
//                      pg_group_exchange_stats; two such steps in a row and the next 64 exchange full lists at once)
//   every shard h      identical deterministic merge → global top-k; compacts the candidates it OWNS, ranks them
//                      (DNN3, embedding rows are local), stores the scores into the slab of the request's TAIL shard
//   tail shard s       owns the requests q = s (mod G): RankScore fusion → ItemRankScore sort → DPP candidates = first
//                      max(page, dpp_candidates) of the sorted list (DPPSort.doSort, sort/dpp_sort.go:280-291)
//   every shard h      stores the embedding rows it owns of every tail shard's candidates into that shard's DPP buffer
//   tail shard s       DPP greedy MAP (sort/dpp_sort.go:372-551) → its requests' pages; one device → host copy each
// Round 2 ran the whole tail on shard 0 while the others idled; spreading it by request divides the ~1.5 ms of
// fusion + sort + DPP per 256 requests by G.
//
// A step is enqueued without waiting for anything (pg_group_recommend_begin) and collected later (_end): each shard
// has TWO lanes — contexts with their own stream and scratch — and consecutive steps alternate between them, so one
// batch's latency-bound tail and its verification run under the next batch's scans, as on a single GPU
// (pipeline.hip).  Each shard's recall plan is verified at _end (a failed plan re-runs the step with that shard's
// fallback plan on the same lane).
#include "pipeline.hpp"

#include <algorithm>

namespace pg {
namespace {

constexpr int kLanes = 2;

// A shard's resources for the steps of one lane
struct Lane {
    pg_ctx* ctx = nullptr;
    PipeRun* run = nullptr;
    bool plan_failed = false;        // the last attempt's recall plan did not hold: the retry runs this shard's next plan
    hipEvent_t ev_lists = nullptr, ev_ready = nullptr, ev_rank = nullptr, ev_sel = nullptr, ev_emb = nullptr, ev_done = nullptr;
    // sized for (nq_cap, k_cap)
    float* d_q = nullptr;
    char* d_own = nullptr;           // this shard's packed block: rows u64 [nq][k] | scores f32 [nq][k]
    char* d_head = nullptr;          // the heads of its lists, packed the same way: rows u64 [nq][m] | scores f32 [nq][m]
    uint32_t* d_tail = nullptr;      // != 0: some shard's last sent entry lies inside a merged top-k (the exchange was too narrow)
    char* g_blk = nullptr;           // G packed blocks, slot i written by shard i
    uint64_t* m_rows = nullptr;      // merged [nq][k]
    float* m_sc = nullptr;
    uint32_t* m_cnt = nullptr;       // [256] valid entries per merged list
    uint32_t* d_local = nullptr;     // compacted local rows of the candidates this shard owns
    uint32_t* d_slot = nullptr;      // their positions q * k + j in the merged lists
    uint32_t* d_off = nullptr;       // [nq + 1] (+ 256 counters)
    float* d_rank = nullptr;         // compacted model scores
    // tail: this shard's requests (q = shard mod G), contiguous
    uint64_t* t_rows = nullptr;      // [nqs][k]
    float* t_recall = nullptr;
    float* t_slab = nullptr;         // model scores, written by the owners
    double* t_fused = nullptr;
    uint32_t* t_order = nullptr;
    uint32_t* t_count = nullptr;     // [256]
    uint32_t* t_err = nullptr;       // [256] RankScore flags
    uint64_t* c_rows = nullptr;      // DPP candidates [nqs][C]
    double* c_rel = nullptr;
    float* c_emb = nullptr;
    uint32_t* c_bail = nullptr;
    uint32_t* pick = nullptr;        // [nqs][top_n]
    uint32_t* pick_cnt = nullptr;    // [256]
    char* d_page = nullptr;
    char* h_page = nullptr;          // pinned
    uint32_t* h_flags = nullptr;     // pinned: [0,256) RankScore flags, [256,512) counts, [512,768) DPP pick counts, [768] the d_tail flag
    // the peers' buffers of the same lane (device arrays of G pointers)
    float** slab_tab = nullptr;
    uint64_t** crows_tab = nullptr;
    float** cemb_tab = nullptr;
};

struct Shard {
    Lane lane[kLanes];
    pg_table* tab = nullptr;
    pg_model* model = nullptr;
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

// the first m entries of every request's list, packed: rows [nq][m] | scores [nq][m] (scores at byte offset sc_off)
__global__ void pack_heads_kernel(const uint64_t* __restrict__ rows, const float* __restrict__ scores, uint32_t nq, uint32_t k, uint32_t m,
                                  uint64_t* __restrict__ h_rows, float* __restrict__ h_scores) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * m) return;
    const uint32_t q = i / m, j = i % m;
    h_rows[i] = rows[(size_t)q * k + j];
    h_scores[i] = scores[(size_t)q * k + j];
}
// after the merge of G lists of m entries: does some list's LAST entry rank strictly inside its request's merged top-k?  Then an
// entry that shard did not send could too.  Order = the recall's (score descending by IEEE totalOrder with NaN last, row
// ascending); a request whose merged list is short of k takes every full list as suspect.  One thread per (list, request).
__device__ __forceinline__ uint32_t tail_ord(float f) {
    const uint32_t b = __float_as_uint(f);
    if (f != f) return 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ void tail_needed_kernel(const char* __restrict__ g_blk, size_t slot_bytes, size_t sc_off, uint32_t G, uint32_t nq, uint32_t m,
                                   const uint64_t* __restrict__ m_rows, const float* __restrict__ m_sc, const uint32_t* __restrict__ m_cnt,
                                   uint32_t k, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * nq) return;
    const uint32_t g = i / nq, q = i % nq;
    const uint64_t r_m = reinterpret_cast<const uint64_t*>(g_blk + (size_t)g * slot_bytes)[(size_t)q * m + m - 1];
    if (r_m == ~0ull) return;                                  // the list ended before its m-th entry: nothing was left unsent
    const float s_m = reinterpret_cast<const float*>(g_blk + (size_t)g * slot_bytes + sc_off)[(size_t)q * m + m - 1];
    bool need = m_cnt[q] < k;
    if (!need) {
        const uint64_t r_k = m_rows[(size_t)q * k + k - 1];
        const uint32_t a = tail_ord(s_m), b = tail_ord(m_sc[(size_t)q * k + k - 1]);
        need = a > b || (a == b && r_m < r_k);
    }
    if (need) atomicOr(flag, 1u);
}

// slab[slot[i]] = score[i] for the owned candidates (slab may live on the lead's device: peer store)
__global__ void scatter_scores_kernel(const float* __restrict__ score, const uint32_t* __restrict__ slot,
                                      const uint32_t* __restrict__ total, uint32_t cap, float* __restrict__ slab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap && i < *total) slab[slot[i]] = score[i];
}

// slab_tab[q % G][(q / G) * k + j] = score[i] for the owned candidates: a request's scores go to its tail shard
// (peer stores when that shard is another device)
__global__ void scatter_to_tails_kernel(const float* __restrict__ score, const uint32_t* __restrict__ slot,
                                        const uint32_t* __restrict__ total, uint32_t cap, uint32_t k, uint32_t G,
                                        float* const* __restrict__ slab_tab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap || i >= *total) return;
    const uint32_t sl = slot[i], q = sl / k, j = sl - q * k;
    slab_tab[q % G][(size_t)(q / G) * k + j] = score[i];
}

// the tail shard's own requests out of the merged lists: t[i] = m[(s + i G)], i < nqs
__global__ void take_requests_kernel(const uint64_t* __restrict__ m_rows, const float* __restrict__ m_sc,
                                     const uint32_t* __restrict__ m_cnt, uint32_t k, uint32_t s, uint32_t G, uint32_t nqs,
                                     uint64_t* __restrict__ t_rows, float* __restrict__ t_sc, uint32_t* __restrict__ t_cnt) {
    const uint32_t i = blockIdx.y;
    const uint32_t q = s + i * G;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < k; j += gridDim.x * blockDim.x) {
        t_rows[(size_t)i * k + j] = m_rows[(size_t)q * k + j];
        t_sc[(size_t)i * k + j] = m_sc[(size_t)q * k + j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) t_cnt[i] = m_cnt[q];
}

// every shard: the embedding rows it owns among the DPP candidates of EVERY tail shard → that shard's buffer
__global__ void gather_to_tails_kernel(const float* __restrict__ tab, uint32_t dim, uint64_t off, uint64_t nrows,
                                       uint64_t* const* __restrict__ crows_tab, float* const* __restrict__ cemb_tab,
                                       uint32_t nq, uint32_t G, uint32_t C) {
    const uint32_t s = blockIdx.y;                              // tail shard
    const uint32_t nqs = s < nq ? (nq - s + G - 1) / G : 0u;
    const uint32_t qpr = dim / 4;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = gid / qpr;
    const uint32_t c = (uint32_t)(gid % qpr);
    if (i >= (uint64_t)nqs * C) return;
    const uint64_t r = crows_tab[s][i];
    if (r == ~0ull || r < off || r - off >= nrows) return;
    *reinterpret_cast<float4*>(cemb_tab[s] + i * dim + 4 * c) = *reinterpret_cast<const float4*>(tab + (r - off) * dim + 4 * c);
}

}  // namespace
}  // namespace pg

struct pg_group_ticket {
    int lane = 0;
    uint32_t nq = 0, k = 0, C = 0, top_n = 0;
    const pg_expr* e = nullptr;
    pg::ExprHold e_hold;
    std::vector<int> src;
    pg_group_plan plan{};
    std::vector<float> users;        // the step's inputs (a failed plan re-enqueues the step)
    bool full_exchange = false;      // exchange the whole lists (the heads were too narrow for this batch, or the group is backing off)
    uint32_t m = 0;                  // entries per request and shard the enqueued attempt exchanged
};

struct pg_group {
    std::vector<pg::Shard> sh;
    std::vector<int> dev;
    uint64_t total_rows = 0;
    uint32_t dim = 0;
    uint32_t nq_cap = 0, k_cap = 0, c_cap = 0, top_cap = 0, nv_cap = 0;
    std::mutex mu;                 // enqueueing / collecting a step, table and model changes
    bool lane_busy[pg::kLanes] = {false, false};
    int next_lane = 0;
    // the first exchange (under mu): steps served, steps repeated with the full lists, the bytes a shard sent last time
    uint64_t ex_steps = 0, ex_round2 = 0, ex_last_bytes = 0, ex_last_entries = 0;
    uint32_t ex_streak = 0, ex_backoff = 0;
};

namespace pg {
namespace {

void shard_range(uint64_t total, uint32_t G, uint32_t g, uint64_t* b, uint64_t* e) {
    const uint64_t base = total / G, rem = total % G;
    *b = g * base + std::min<uint64_t>(g, rem);
    *e = *b + base + (g < rem ? 1 : 0);
}

size_t blk_rows_bytes(size_t n) { return (n * 8 + 255) & ~(size_t)255; }
size_t blk_bytes(size_t n) { return blk_rows_bytes(n) + ((n * 4 + 255) & ~(size_t)255); }

void free_lane_buffers(Lane& l) {
    if (!l.ctx) return;
    hipSetDevice(l.ctx->device);
    for (void* p : {(void*)l.d_q, (void*)l.d_own, (void*)l.d_head, (void*)l.d_tail, (void*)l.g_blk, (void*)l.m_rows, (void*)l.m_sc, (void*)l.m_cnt, (void*)l.d_local,
                    (void*)l.d_slot, (void*)l.d_off, (void*)l.d_rank, (void*)l.t_rows, (void*)l.t_recall, (void*)l.t_slab, (void*)l.t_fused,
                    (void*)l.t_order, (void*)l.t_count, (void*)l.t_err, (void*)l.c_rows, (void*)l.c_rel, (void*)l.c_emb, (void*)l.c_bail,
                    (void*)l.pick, (void*)l.pick_cnt, (void*)l.d_page, (void*)l.slab_tab, (void*)l.crows_tab, (void*)l.cemb_tab})
        if (p) hipFree(p);
    if (l.h_page) hipHostFree(l.h_page);
    if (l.h_flags) hipHostFree(l.h_flags);
    l.d_q = nullptr; l.d_own = nullptr; l.d_head = nullptr; l.d_tail = nullptr; l.g_blk = nullptr; l.m_rows = nullptr; l.m_sc = nullptr; l.m_cnt = nullptr; l.d_local = nullptr;
    l.d_slot = nullptr; l.d_off = nullptr; l.d_rank = nullptr; l.t_rows = nullptr; l.t_recall = nullptr; l.t_slab = nullptr;
    l.t_fused = nullptr; l.t_order = nullptr; l.t_count = nullptr; l.t_err = nullptr; l.c_rows = nullptr; l.c_rel = nullptr;
    l.c_emb = nullptr; l.c_bail = nullptr; l.pick = nullptr; l.pick_cnt = nullptr; l.d_page = nullptr; l.h_page = nullptr;
    l.h_flags = nullptr; l.slab_tab = nullptr; l.crows_tab = nullptr; l.cemb_tab = nullptr;
}

void free_step_buffers(pg_group* g) {
    for (auto& s : g->sh)
        for (auto& l : s.lane) free_lane_buffers(l);
    g->nq_cap = g->k_cap = g->c_cap = g->top_cap = g->nv_cap = 0;
}

int ensure_step_buffers(pg_group* g, uint32_t nq, uint32_t k, uint32_t C, uint32_t top_n, int nv) {
    if (nq <= g->nq_cap && k <= g->k_cap && C <= g->c_cap && top_n <= g->top_cap && (uint32_t)nv <= g->nv_cap) return PG_OK;
    if (g->lane_busy[0] || g->lane_busy[1]) {
        set_error("pg_group: a step with larger shapes than any before cannot start while another one is outstanding (end it first)");
        return PG_ERR_INVALID;
    }
    for (auto& s : g->sh)
        for (auto& l : s.lane) {
            hipSetDevice(l.ctx->device);
            hipStreamSynchronize(l.ctx->stream);
        }
    // grow every dimension to the largest seen so far: alternating shapes must not re-allocate on every call, and a
    // later expression with more variables must not find a buffer sized for the first one's
    nq = std::max(nq, g->nq_cap); k = std::max(k, g->k_cap); C = std::max(C, g->c_cap); top_n = std::max(top_n, g->top_cap);
    nv = std::max(nv, (int)g->nv_cap);
    free_step_buffers(g);
    const uint32_t G = (uint32_t)g->sh.size();
    const size_t n = (size_t)nq * k;
    const uint32_t nqs = (nq + G - 1) / G;                  // requests of one tail shard, at most
    const size_t ns = (size_t)nqs * k;
    const size_t C1 = std::max(C, 1u);
    const size_t page = (size_t)nqs * top_n * page_entry_bytes(1);
    for (auto& s : g->sh)
        for (auto& l : s.lane) {
            PG_HIP(hipSetDevice(l.ctx->device));
            PG_HIP(hipMalloc((void**)&l.d_q, (size_t)nq * g->dim * 4));
            PG_HIP(hipMalloc((void**)&l.d_own, blk_bytes(n)));
            PG_HIP(hipMalloc((void**)&l.g_blk, blk_bytes(n) * G));
            PG_HIP(hipMalloc((void**)&l.d_head, blk_bytes(n)));
            PG_HIP(hipMalloc((void**)&l.d_tail, 256));
            PG_HIP(hipMalloc((void**)&l.m_rows, n * 8));
            PG_HIP(hipMalloc((void**)&l.m_sc, n * 4));
            PG_HIP(hipMalloc((void**)&l.m_cnt, 256 * 4));
            PG_HIP(hipMalloc((void**)&l.d_local, n * 4));
            PG_HIP(hipMalloc((void**)&l.d_slot, n * 4));
            PG_HIP(hipMalloc((void**)&l.d_off, ((size_t)nq + 1 + 256) * 4));
            PG_HIP(hipMalloc((void**)&l.d_rank, n * 4));
            PG_HIP(hipMalloc((void**)&l.t_rows, ns * 8));
            PG_HIP(hipMalloc((void**)&l.t_recall, ns * 4));
            PG_HIP(hipMalloc((void**)&l.t_slab, ns * 4));
            PG_HIP(hipMalloc((void**)&l.t_fused, ns * 8));
            PG_HIP(hipMalloc((void**)&l.t_order, ns * 4));
            PG_HIP(hipMalloc((void**)&l.t_count, 256 * 4));
            PG_HIP(hipMalloc((void**)&l.t_err, 256 * 4));
            PG_HIP(hipMalloc((void**)&l.c_rows, (size_t)nqs * C1 * 8));
            PG_HIP(hipMalloc((void**)&l.c_rel, (size_t)nqs * C1 * 8));
            PG_HIP(hipMalloc((void**)&l.c_emb, (size_t)nqs * C1 * g->dim * 4));
            PG_HIP(hipMalloc((void**)&l.c_bail, 256 * 4));
            PG_HIP(hipMalloc((void**)&l.pick, (size_t)nqs * top_n * 4));
            PG_HIP(hipMalloc((void**)&l.pick_cnt, 256 * 4));
            PG_HIP(hipMalloc((void**)&l.d_page, page));
            PG_HIP(hipHostMalloc((void**)&l.h_page, page));
            PG_HIP(hipHostMalloc((void**)&l.h_flags, 772 * 4));
            PG_HIP(hipMalloc((void**)&l.slab_tab, (size_t)G * sizeof(void*)));
            PG_HIP(hipMalloc((void**)&l.crows_tab, (size_t)G * sizeof(void*)));
            PG_HIP(hipMalloc((void**)&l.cemb_tab, (size_t)G * sizeof(void*)));
        }
    // every shard's table of its peers' buffers, per lane
    for (int L = 0; L < kLanes; ++L) {
        std::vector<float*> slabs(G), cembs(G);
        std::vector<uint64_t*> crows(G);
        for (uint32_t i = 0; i < G; ++i) {
            slabs[i] = g->sh[i].lane[L].t_slab;
            cembs[i] = g->sh[i].lane[L].c_emb;
            crows[i] = g->sh[i].lane[L].c_rows;
        }
        for (auto& s : g->sh) {
            Lane& l = s.lane[L];
            PG_HIP(hipSetDevice(l.ctx->device));
            PG_HIP(hipMemcpy(l.slab_tab, slabs.data(), (size_t)G * sizeof(void*), hipMemcpyHostToDevice));
            PG_HIP(hipMemcpy(l.cemb_tab, cembs.data(), (size_t)G * sizeof(void*), hipMemcpyHostToDevice));
            PG_HIP(hipMemcpy(l.crows_tab, crows.data(), (size_t)G * sizeof(void*), hipMemcpyHostToDevice));
        }
    }
    g->nq_cap = nq; g->k_cap = k; g->c_cap = C; g->top_cap = top_n; g->nv_cap = (uint32_t)std::max(nv, 1);
    return PG_OK;
}

// the tail shard's call record over lane l's buffers (requests [0, nqs) of that shard)
void tail_call(const pg_group* g, const Shard& s, const Lane& l, const pg_group_ticket* tk, uint32_t nqs, RecommendCall* c) {
    *c = RecommendCall();
    c->t = s.tab;
    c->n_algos = 1;
    c->algos[0].m = s.model;
    c->e = tk->e;
    c->var_src = tk->src.data();
    c->nv = pg_expr_num_vars(tk->e);
    c->nq = nqs;
    c->k = tk->k;
    c->d_rows = l.t_rows;
    c->d_recall = l.t_recall;
    c->d_rank = l.t_slab;
    c->rank_stride = (size_t)nqs * tk->k;
    c->d_fused = l.t_fused;
    c->d_order = l.t_order;
    c->d_count = l.t_count;
    c->pads = false;
    c->rerank.kind = tk->C ? 1 : 0;
    c->rerank.candidates = tk->C;
    c->rerank.dpp.alpha = tk->plan.dpp_alpha;
    c->rerank.dpp.window = tk->plan.dpp_window;
    c->rerank.dpp.normalize_emb = tk->plan.dpp_normalize_emb;
    c->rerank.dpp.norm_relevance_score = 0;
    c->top_n = tk->top_n;
    c->d_pick = l.pick;
    c->d_pick_cnt = l.pick_cnt;
    (void)g;
}

// enqueue one attempt of the step on the ticket's lane; nothing waits for the device
int step_enqueue(pg_group* g, pg_group_ticket* tk, uint32_t attempt) {
    const uint32_t G = (uint32_t)g->sh.size(), nq = tk->nq, k = tk->k, C = tk->C, top_n = tk->top_n;
    const uint32_t n = nq * k;
    const int L = tk->lane;
    const size_t bb = blk_bytes(n), rb = blk_rows_bytes(n);
    int rc;
    // ---- 1. local recall on every shard ---------------------------------------------------------------------------
    for (uint32_t i = 0; i < G; ++i) {
        Shard& s = g->sh[i];
        Lane& l = s.lane[L];
        PG_HIP(hipSetDevice(l.ctx->device));
        PG_HIP(hipMemcpyAsync(l.d_q, tk->users.data(), (size_t)nq * g->dim * 4, hipMemcpyHostToDevice, l.ctx->stream));
        std::lock_guard<std::mutex> cg(l.ctx->mu);
        RecallJob& j = l.run->job;
        const bool retry_next_plan = attempt > 0 && l.plan_failed && j.table_gen == s.tab->generation.load(std::memory_order_relaxed);
        if (!retry_next_plan) {
            j = RecallJob();
            j.ctx = l.ctx;
            j.t = s.tab;
            j.d_queries = l.d_q;
            j.nq = nq;
            j.k = k;
            j.d_out_rows = (uint64_t*)l.d_own;
            j.d_out_scores = (float*)(l.d_own + rb);
            j.h_status = l.run->h_status;
            j.events = &l.run->events;
            if ((rc = recall_job_prepare(&j))) return rc;
        }
        if ((rc = recall_job_enqueue(&j))) return rc;
    }
    // ---- 2. all-gather: shard i stores the heads of its lists (or, for a repeated / backed-off step, the whole lists) into slot i
    //         of every shard — one copy per peer; every tail shard clears its slab (padding slots have no owner) ------------
    uint32_t m = k;
    if (!tk->full_exchange && G > 1) {
        const double per = (double)k / G;
        const uint32_t w = (uint32_t)ceil(per + 6.0 * sqrt(per) + 8.0);
        if (w < k) m = w;
    }
    tk->m = m;
    const size_t nm = (size_t)nq * m;
    const size_t hb = blk_bytes(nm), hrb = blk_rows_bytes(nm);      // bytes of one packed block of heads, of its rows part
    for (uint32_t i = 0; i < G; ++i) {
        Lane& l = g->sh[i].lane[L];
        PG_HIP(hipSetDevice(l.ctx->device));
        hipStream_t st = l.ctx->stream;
        const char* src = l.d_own;
        if (m < k) {
            pack_heads_kernel<<<(uint32_t)((nm + 255) / 256), 256, 0, st>>>((const uint64_t*)l.d_own, (const float*)(l.d_own + rb), nq, k, m,
                                                                            (uint64_t*)l.d_head, (float*)(l.d_head + hrb));
            PG_HIP(hipGetLastError());
            src = l.d_head;
        }
        for (uint32_t h = 0; h < G; ++h)
            PG_HIP(hipMemcpyAsync(g->sh[h].lane[L].g_blk + (size_t)i * bb, src, m < k ? hb : bb, hipMemcpyDeviceToDevice, st));
        PG_HIP(hipEventRecord(l.ev_lists, st));
        const uint32_t nqs = i < nq ? (nq - i + G - 1) / G : 0u;
        if (nqs) PG_HIP(hipMemsetAsync(l.t_slab, 0, (size_t)nqs * k * 4, st));
        PG_HIP(hipMemsetAsync(l.d_tail, 0, 4, st));
        PG_HIP(hipEventRecord(l.ev_ready, st));
    }
    // ---- 3. every shard: merge, rank what it owns, store the scores into the tail shards' slabs --------------------
    for (uint32_t h = 0; h < G; ++h) {
        Shard& s = g->sh[h];
        Lane& l = s.lane[L];
        PG_HIP(hipSetDevice(l.ctx->device));
        hipStream_t st = l.ctx->stream;
        for (uint32_t i = 0; i < G; ++i)
            if (i != h) PG_HIP(hipStreamWaitEvent(st, g->sh[i].lane[L].ev_lists, 0));
        {
            std::lock_guard<std::mutex> cg(l.ctx->mu);
            // (slot stride bb whatever was sent; inside a slot the block is packed for m entries per request)
            if ((rc = topk_merge_strided_locked(l.ctx, (const uint64_t*)l.g_blk, (const float*)(l.g_blk + (m < k ? hrb : rb)), nq, G, m, bb / 8, m,
                                                bb / 4, m, k, l.m_rows, l.m_sc, l.m_cnt)))
                return rc;
            if (m < k) {
                tail_needed_kernel<<<(G * nq + 255) / 256, 256, 0, st>>>(l.g_blk, bb, hrb, G, nq, m, l.m_rows, l.m_sc, l.m_cnt, k, l.d_tail);
                PG_HIP(hipGetLastError());
            }
            uint32_t* d_cnt = l.d_off + nq + 1;
            owned_count_kernel<<<nq, 256, 0, st>>>(l.m_rows, k, s.tab->row_offset, s.tab->rows, d_cnt);
            owned_scan_kernel<<<1, 256, 0, st>>>(d_cnt, nq, l.d_off);
            owned_fill_kernel<<<nq, 1024, 0, st>>>(l.m_rows, k, s.tab->row_offset, s.tab->rows, l.d_off, l.d_local, l.d_slot);
            PG_HIP(hipGetLastError());
            if ((rc = rank_dnn3_dev_locked(l.ctx, s.model, s.tab, l.d_q, l.d_local, l.d_off, nq, n, l.d_rank))) return rc;
        }
        for (uint32_t i = 0; i < G; ++i)
            if (i != h) PG_HIP(hipStreamWaitEvent(st, g->sh[i].lane[L].ev_ready, 0));
        scatter_to_tails_kernel<<<(n + 255) / 256, 256, 0, st>>>(l.d_rank, l.d_slot, l.d_off + nq, n, k, G, l.slab_tab);
        PG_HIP(hipGetLastError());
        PG_HIP(hipEventRecord(l.ev_rank, st));
    }
    // ---- 4. every tail shard: its requests' fusion, sort, DPP candidates ----------------------------------------------
    std::vector<RecommendCall> calls(G);
    std::vector<PostScratch> pss(G);
    for (uint32_t s_ = 0; s_ < G; ++s_) {
        Shard& s = g->sh[s_];
        Lane& l = s.lane[L];
        const uint32_t nqs = s_ < nq ? (nq - s_ + G - 1) / G : 0u;
        PG_HIP(hipSetDevice(l.ctx->device));
        hipStream_t st = l.ctx->stream;
        if (nqs) {
            for (uint32_t h = 0; h < G; ++h)
                if (h != s_) PG_HIP(hipStreamWaitEvent(st, g->sh[h].lane[L].ev_rank, 0));
            take_requests_kernel<<<dim3((k + 255) / 256, nqs), 256, 0, st>>>(l.m_rows, l.m_sc, l.m_cnt, k, s_, G, nqs, l.t_rows, l.t_recall,
                                                                             l.t_count);
            PG_HIP(hipGetLastError());
            tail_call(g, s, l, tk, nqs, &calls[s_]);
            std::lock_guard<std::mutex> cg(l.ctx->mu);
            if ((rc = post_scratch(l.ctx, calls[s_], nqs, &pss[s_]))) return rc;
            // (the re-rank buffers are the lane's own: peers store embeddings into them)
            pss[s_].c_rows = l.c_rows;
            pss[s_].c_rel = l.c_rel;
            pss[s_].c_emb = l.c_emb;
            pss[s_].c_bail = l.c_bail;
            if ((rc = uniform_offsets_locked(l.ctx, nqs, k, pss[s_].d_off))) return rc;
            if ((rc = post_fuse_sort_locked(l.ctx, calls[s_], 0, nqs, pss[s_]))) return rc;
            PG_HIP(hipMemcpyAsync(l.t_err, pss[s_].d_err, (size_t)nqs * 4, hipMemcpyDeviceToDevice, st));
            if (C && (rc = rerank_select_locked(l.ctx, calls[s_], 0, nqs, pss[s_]))) return rc;
        }
        if (C) PG_HIP(hipEventRecord(l.ev_sel, st));
    }
    if (C) {
        // ---- 5. every shard stores the candidate embeddings it owns into the tail shards' DPP buffers ------------------
        const uint32_t nqs_max = (nq + G - 1) / G;
        for (uint32_t h = 0; h < G; ++h) {
            Shard& s = g->sh[h];
            Lane& l = s.lane[L];
            PG_HIP(hipSetDevice(l.ctx->device));
            hipStream_t st = l.ctx->stream;
            for (uint32_t i = 0; i < G; ++i)
                if (i != h) PG_HIP(hipStreamWaitEvent(st, g->sh[i].lane[L].ev_sel, 0));
            const uint64_t threads = (uint64_t)nqs_max * C * (g->dim / 4);
            gather_to_tails_kernel<<<dim3((uint32_t)((threads + 255) / 256), G), 256, 0, st>>>(s.tab->d, g->dim, s.tab->row_offset, s.tab->rows,
                                                                                            l.crows_tab, l.cemb_tab, nq, G, C);
            PG_HIP(hipGetLastError());
            PG_HIP(hipEventRecord(l.ev_emb, st));
        }
        for (uint32_t s_ = 0; s_ < G; ++s_) {
            Lane& l = g->sh[s_].lane[L];
            const uint32_t nqs = s_ < nq ? (nq - s_ + G - 1) / G : 0u;
            if (!nqs) continue;
            PG_HIP(hipSetDevice(l.ctx->device));
            for (uint32_t h = 0; h < G; ++h)
                if (h != s_) PG_HIP(hipStreamWaitEvent(l.ctx->stream, g->sh[h].lane[L].ev_emb, 0));
            std::lock_guard<std::mutex> cg(l.ctx->mu);
            if ((rc = rerank_run_locked(l.ctx, calls[s_], 0, nqs, pss[s_]))) return rc;
        }
    }
    // ---- 6. the pages, one device → host copy per tail shard ---------------------------------------------------------------
    for (uint32_t s_ = 0; s_ < G; ++s_) {
        Lane& l = g->sh[s_].lane[L];
        const uint32_t nqs = s_ < nq ? (nq - s_ + G - 1) / G : 0u;
        PG_HIP(hipSetDevice(l.ctx->device));
        hipStream_t st = l.ctx->stream;
        if (nqs) {
            if ((rc = page_launch(st, l.t_order, C ? l.pick : nullptr, l.pick_cnt, l.t_rows, l.t_recall, l.t_slab, (size_t)nqs * k, 1, l.t_fused,
                                  nqs, k, top_n, l.d_page)))
                return rc;
            PG_HIP(hipMemcpyAsync(l.h_page, l.d_page, (size_t)nqs * top_n * page_entry_bytes(1), hipMemcpyDeviceToHost, st));
            PG_HIP(hipMemcpyAsync(l.h_flags, l.t_err, (size_t)nqs * 4, hipMemcpyDeviceToHost, st));
            PG_HIP(hipMemcpyAsync(l.h_flags + 256, l.t_count, (size_t)nqs * 4, hipMemcpyDeviceToHost, st));
            if (C) PG_HIP(hipMemcpyAsync(l.h_flags + 512, l.pick_cnt, (size_t)nqs * 4, hipMemcpyDeviceToHost, st));
        }
        PG_HIP(hipMemcpyAsync(l.h_flags + 768, l.d_tail, 4, hipMemcpyDeviceToHost, st));
        PG_HIP(hipEventRecord(l.ev_done, st));
    }
    return PG_OK;
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

int pg_dpp_kernel_matrix_dev(pg_ctx* ctx, const float* d_emb, const double* d_rel, uint32_t n_req, uint32_t n, uint32_t dim,
                             double alpha, int normalize_emb, double* d_out_L) {
    PG_REQUIRE(ctx && d_emb && d_rel && d_out_L, "pg_dpp_kernel_matrix_dev: NULL argument");
    PG_REQUIRE(n <= 8192 && dim <= 4096 && n_req <= 65535, "pg_dpp_kernel_matrix_dev: %u requests x %u candidates x %u dims unsupported", n_req, n, dim);
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::dpp_run_locked(ctx, d_emb, nullptr, d_rel, n_req, n, dim, 0, alpha, 0, 0, normalize_emb, 1, 1, nullptr, nullptr, d_out_L);
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
    for (uint32_t i = 0; i < n_shards && !rc; ++i)
        for (int L = 0; L < pg::kLanes && !rc; ++L) {
            pg::Lane& l = g->sh[i].lane[L];
            rc = pg_init(devices[i], nullptr, &l.ctx);
            if (rc) break;
            for (hipEvent_t* e : {&l.ev_lists, &l.ev_ready, &l.ev_rank, &l.ev_sel, &l.ev_emb, &l.ev_done})
                if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
                    pg::set_error("pg_group_create: %s", hipGetErrorString(hipGetLastError()));
                    rc = PG_ERR_DEVICE;
                    break;
                }
            if (!rc) rc = pg::pipe_run_acquire(l.ctx, &l.run);
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
        for (auto& l : s.lane)
            if (l.ctx) {
                hipSetDevice(l.ctx->device);
                hipStreamSynchronize(l.ctx->stream);
            }
    pg::free_step_buffers(g);
    for (auto& s : g->sh) {
        pg_ctx* c0 = s.lane[0].ctx;
        if (c0) {
            hipSetDevice(c0->device);
            if (s.model) pg_model_destroy(c0, s.model);
            if (s.tab) pg_table_destroy(c0, s.tab);
        }
        for (auto& l : s.lane) {
            if (!l.ctx) continue;
            for (hipEvent_t e : {l.ev_lists, l.ev_ready, l.ev_rank, l.ev_sel, l.ev_emb, l.ev_done})
                if (e) hipEventDestroy(e);
            if (l.run) pg::pipe_run_release(l.ctx, l.run);
            pg_shutdown(l.ctx);
        }
    }
    delete g;
    return PG_OK;
}

uint32_t pg_group_size(const pg_group* g) { return g ? (uint32_t)g->sh.size() : 0; }
int pg_group_info(const pg_group* g, uint64_t* total_rows, uint32_t* dim) {
    PG_REQUIRE(g, "pg_group_info: NULL argument");
    if (total_rows) *total_rows = g->total_rows;
    if (dim) *dim = g->dim;
    return PG_OK;
}
pg_ctx* pg_group_ctx(pg_group* g, uint32_t shard) { return g && shard < g->sh.size() ? g->sh[shard].lane[0].ctx : nullptr; }
pg_table* pg_group_table(pg_group* g, uint32_t shard) { return g && shard < g->sh.size() ? g->sh[shard].tab : nullptr; }

static int group_quiesce(pg_group* g, const char* who) {
    if (g->lane_busy[0] || g->lane_busy[1]) {
        pg::set_error("%s: a step is outstanding (end its ticket first)", who);
        return PG_ERR_INVALID;
    }
    return PG_OK;
}

int pg_group_table_create(pg_group* g, uint64_t total_rows, uint32_t dim) {
    PG_REQUIRE(g && total_rows >= g->sh.size(), "pg_group_table_create: bad argument");
    PG_REQUIRE(total_rows < 0xFFFFFFFFull, "pg_group_table_create: global row ids must stay below 2^32");
    std::lock_guard<std::mutex> lk(g->mu);
    int rc;
    if ((rc = group_quiesce(g, "pg_group_table_create"))) return rc;
    const uint32_t G = (uint32_t)g->sh.size();
    for (uint32_t i = 0; i < G; ++i) {
        auto& s = g->sh[i];
        pg_ctx* c0 = s.lane[0].ctx;
        if (s.tab) {
            for (auto& l : s.lane) pg_synchronize(l.ctx);
            pg_table_destroy(c0, s.tab);
            s.tab = nullptr;
        }
        uint64_t b, e;
        pg::shard_range(total_rows, G, i, &b, &e);
        if ((rc = pg_table_create(c0, e - b, dim, b, &s.tab))) return rc;
    }
    g->total_rows = total_rows;
    g->dim = dim;
    return PG_OK;
}

int pg_group_table_fill_synthetic(pg_group* g, uint64_t seed, int normalize) {
    PG_REQUIRE(g && g->total_rows, "pg_group_table_fill_synthetic: no table");
    std::lock_guard<std::mutex> lk(g->mu);
    int rc;
    if ((rc = group_quiesce(g, "pg_group_table_fill_synthetic"))) return rc;
    for (auto& s : g->sh)
        if ((rc = pg_table_fill_synthetic(s.lane[0].ctx, s.tab, seed, normalize))) return rc;
    return PG_OK;
}

int pg_group_table_upload(pg_group* g, uint64_t row0, uint64_t nrows, const float* host_rows) {
    PG_REQUIRE(g && g->total_rows && (nrows == 0 || host_rows), "pg_group_table_upload: bad argument");
    PG_REQUIRE(row0 + nrows <= g->total_rows, "pg_group_table_upload: rows %llu..%llu outside the table",
               (unsigned long long)row0, (unsigned long long)(row0 + nrows));
    std::lock_guard<std::mutex> lk(g->mu);
    int rc;
    if ((rc = group_quiesce(g, "pg_group_table_upload"))) return rc;
    const uint32_t G = (uint32_t)g->sh.size();
    for (uint32_t i = 0; i < G; ++i) {
        uint64_t b, e;
        pg::shard_range(g->total_rows, G, i, &b, &e);
        const uint64_t lo = std::max(b, row0), hi = std::min(e, row0 + nrows);
        if (lo >= hi) continue;
        if ((rc = pg_table_upload(g->sh[i].lane[0].ctx, g->sh[i].tab, lo - b, hi - lo, host_rows + (lo - row0) * g->dim))) return rc;
    }
    return PG_OK;
}

int pg_group_model_load(pg_group* g, pg_model_kind kind, pg_prec prec, const void* blob, size_t len) {
    PG_REQUIRE(g && blob, "pg_group_model_load: NULL argument");
    std::lock_guard<std::mutex> lk(g->mu);
    int rc;
    if ((rc = group_quiesce(g, "pg_group_model_load"))) return rc;
    for (auto& s : g->sh) {                      // weights are replicated (0.5 MB)
        pg_ctx* c0 = s.lane[0].ctx;
        if (s.model) {
            for (auto& l : s.lane) pg_synchronize(l.ctx);
            pg_model_destroy(c0, s.model);
            s.model = nullptr;
        }
        if ((rc = pg_model_load(c0, kind, prec, blob, len, &s.model))) return rc;
    }
    return PG_OK;
}

int pg_group_recommend_begin(pg_group* g, const pg_expr* e, const char* rank_var, const pg_group_plan* plan,
                             const float* user_vecs, uint32_t nq, uint32_t top_n, pg_group_ticket** out) {
    PG_REQUIRE(g && e && rank_var && plan && user_vecs && out, "pg_group_recommend: NULL argument");
    PG_REQUIRE(g->total_rows && g->sh[0].model, "pg_group_recommend: the group has no table or no model");
    const uint32_t k = plan->k, G = (uint32_t)g->sh.size();
    PG_REQUIRE(nq >= 1 && nq <= (uint32_t)pg::kMaxQueries && k >= 1 && k <= 16384, "pg_group_recommend: bad nq / k");
    PG_REQUIRE(top_n >= 1 && top_n <= k, "pg_group_recommend: top_n %u outside 1..k", top_n);
    PG_REQUIRE(g->total_rows >= k, "pg_group_recommend: the table has fewer rows than k");
    PG_REQUIRE((uint64_t)G * k <= (uint64_t)k + (1u << 19), "pg_group_recommend: %u shards x k=%u exceeds the merge capacity", G, k);
    const pg_model* m0 = g->sh[0].model;
    PG_REQUIRE(m0->kind == PG_MODEL_DNN3 && m0->d_item == g->dim && m0->d_user == g->dim,
               "pg_group_recommend: the model must be DNN3 with d_user = d_item = the table's dim");
    PG_REQUIRE(m0->n_out == 1, "pg_group_recommend: multi-output models are served by a scene coalescer on one GPU (pg_coalescer_create_scene)");
    const uint32_t C = plan->dpp_candidates ? std::min(k, std::max(top_n, plan->dpp_candidates)) : 0u;
    PG_REQUIRE(C <= 8192, "pg_group_recommend: %u DPP candidates (at most 8192)", C);
    pg_group_ticket* tk = new pg_group_ticket();
    tk->e_hold.take(e);
    int rc;
    if ((rc = pg::recommend_bind_vars(e, &rank_var, 1, &tk->src, "pg_group_recommend"))) {
        delete tk;
        return rc;
    }
    tk->nq = nq; tk->k = k; tk->C = C; tk->top_n = top_n; tk->e = e; tk->plan = *plan;
    tk->users.assign(user_vecs, user_vecs + (size_t)nq * g->dim);
    std::lock_guard<std::mutex> lk(g->mu);
    if ((rc = pg::ensure_step_buffers(g, nq, k, C, top_n, (int)tk->src.size()))) {
        delete tk;
        return rc;
    }
    int L = g->next_lane;
    if (g->lane_busy[L]) L ^= 1;
    if (g->lane_busy[L]) {
        pg::set_error("pg_group_recommend_begin: %d steps are outstanding already (end one first)", pg::kLanes);
        delete tk;
        return PG_ERR_INVALID;
    }
    tk->lane = L;
    if (g->ex_backoff) {                              // (the heads kept proving too narrow: whole lists for now)
        g->ex_backoff--;
        tk->full_exchange = true;
        if (!g->ex_backoff) g->ex_streak = 0;
    }
    for (auto& s : g->sh) s.lane[L].plan_failed = false;
    if ((rc = pg::step_enqueue(g, tk, 0))) {
        for (auto& s : g->sh) {                       // leave nothing half-enqueued behind
            hipSetDevice(s.lane[L].ctx->device);
            hipStreamSynchronize(s.lane[L].ctx->stream);
        }
        delete tk;
        return rc;
    }
    g->lane_busy[L] = true;
    g->next_lane = L ^ 1;
    *out = tk;
    return PG_OK;
}

int pg_group_recommend_end(pg_group* g, pg_group_ticket* tk, uint64_t* out_rows, float* out_recall_scores,
                           float* out_rank_scores, double* out_fused, uint32_t* out_count) {
    PG_REQUIRE(g && tk && out_rows && out_recall_scores && out_rank_scores && out_fused, "pg_group_recommend_end: NULL argument");
    const uint32_t G = (uint32_t)g->sh.size(), nq = tk->nq, top_n = tk->top_n, C = tk->C;
    const int L = tk->lane;
    int rc = PG_OK;
    for (uint32_t attempt = 0; !rc; ++attempt) {
        // the step's completion: every tail shard's page copy (nothing else of the lane runs behind them)
        for (uint32_t i = 0; i < G && !rc; ++i) {
            pg::Lane& l = g->sh[i].lane[L];
            hipSetDevice(l.ctx->device);
            if (hipEventSynchronize(l.ev_done) != hipSuccess || hipStreamSynchronize(l.ctx->stream) != hipSuccess) {
                pg::set_error("pg_group_recommend_end: %s", hipGetErrorString(hipGetLastError()));
                rc = PG_ERR_DEVICE;
            }
        }
        if (rc) break;
        // the deferred verification of every shard's recall plan
        bool all_ok = true;
        for (uint32_t i = 0; i < G && !rc; ++i) {
            pg::Lane& l = g->sh[i].lane[L];
            std::lock_guard<std::mutex> cg(l.ctx->mu);
            bool ok = false;
            if ((rc = pg::recall_job_check(&l.run->job, &ok))) break;
            l.plan_failed = !ok;                              // remembered for the retry: this shard moves to its next plan
            if (ok) pg::recall_job_finish(&l.run->job);
            all_ok = all_ok && ok;
        }
        // ... and of the first exchange: a shard's last sent entry inside a merged top-k means its unsent tail could matter —
        // the step runs again with the whole lists (every shard computed the flag from the same bytes)
        bool narrow = false;
        if (!rc && all_ok && !tk->full_exchange) {
            for (uint32_t i = 0; i < G; ++i) narrow = narrow || g->sh[i].lane[L].h_flags[768] != 0;
        }
        {
            std::lock_guard<std::mutex> lk(g->mu);
            if (!rc && all_ok) {
                if (narrow) {
                    g->ex_round2++;
                    if (++g->ex_streak >= 2) g->ex_backoff = 64;     // (rows in score order, one shard holding the answers: full lists at once for a while)
                } else {
                    g->ex_steps++;
                    g->ex_last_entries = tk->m;
                    g->ex_last_bytes = (uint64_t)nq * tk->m * 12;
                    if (!tk->full_exchange) g->ex_streak = 0;
                }
            }
        }
        if (narrow) {
            tk->full_exchange = true;
            all_ok = false;
        }
        if (rc || all_ok) break;
        // a shard's job runs out of plans by itself (recall_job_enqueue: "overflow in safe mode"); with the threshold model's
        // plan in front and one restart on the exact scan after a screened overflow, a valid sequence takes up to
        // 2 x (plans) enqueues — the bound here only stops a loop that makes no progress
        if (attempt >= 12) {
            pg::set_error("pg_group_recommend: recall plans kept failing (internal error)");
            rc = PG_ERR_DEVICE;
            break;
        }
        std::lock_guard<std::mutex> lk(g->mu);
        rc = pg::step_enqueue(g, tk, attempt + 1);
    }
    if (!rc) {
        // reassemble: tail shard s holds the pages of requests s, s + G, ...
        for (uint32_t s_ = 0; s_ < G && !rc; ++s_) {
            const pg::Lane& l = g->sh[s_].lane[L];
            const uint32_t nqs = s_ < nq ? (nq - s_ + G - 1) / G : 0u;
            const size_t np = (size_t)nqs * top_n;
            const uint64_t* p_rows = (const uint64_t*)l.h_page;
            const double* p_fused = (const double*)(p_rows + np);
            const float* p_recall = (const float*)(p_fused + np);
            const float* p_rank = p_recall + np;
            for (uint32_t i = 0; i < nqs; ++i) {
                const uint32_t q = s_ + i * G;
                if (l.h_flags[i]) {
                    pg::set_expr_arith_error(tk->e);
                    rc = PG_ERR_ARITH;
                    break;
                }
                memcpy(out_rows + (size_t)q * top_n, p_rows + (size_t)i * top_n, (size_t)top_n * 8);
                memcpy(out_fused + (size_t)q * top_n, p_fused + (size_t)i * top_n, (size_t)top_n * 8);
                memcpy(out_recall_scores + (size_t)q * top_n, p_recall + (size_t)i * top_n, (size_t)top_n * 4);
                memcpy(out_rank_scores + (size_t)q * top_n, p_rank + (size_t)i * top_n, (size_t)top_n * 4);
                if (out_count) out_count[q] = C ? l.h_flags[512 + i] : std::min(top_n, l.h_flags[256 + i]);
            }
        }
    }
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->lane_busy[L] = false;
    }
    delete tk;
    return rc;
}

int pg_group_exchange_stats(pg_group* g, uint64_t* out4) {
    PG_REQUIRE(g && out4, "pg_group_exchange_stats: NULL argument");
    std::lock_guard<std::mutex> lk(g->mu);
    out4[0] = g->ex_steps;
    out4[1] = g->ex_round2;
    out4[2] = g->ex_last_bytes;
    out4[3] = g->ex_last_entries;
    return PG_OK;
}

int pg_group_recommend(pg_group* g, const pg_expr* e, const char* rank_var, const pg_group_plan* plan,
                       const float* user_vecs, uint32_t nq, uint32_t top_n, uint64_t* out_rows,
                       float* out_recall_scores, float* out_rank_scores, double* out_fused, uint32_t* out_count) {
    PG_REQUIRE(out_rows && out_recall_scores && out_rank_scores && out_fused, "pg_group_recommend: NULL argument");
    pg_group_ticket* tk = nullptr;
    int rc;
    if ((rc = pg_group_recommend_begin(g, e, rank_var, plan, user_vecs, nq, top_n, &tk))) return rc;
    return pg_group_recommend_end(g, tk, out_rows, out_recall_scores, out_rank_scores, out_fused, out_count);
}

}  // extern "C"
