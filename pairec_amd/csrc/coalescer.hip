// coalescer.hip — cross-request batching inside the library (SURVEY.md 8b "Threading").
//
// pairec calls its plug-ins per request and concurrently: RecallService.GetItems starts one goroutine per recall
// (service/recall.go:129-145), RankService.Rank one per 100-item batch and algorithm (service/rank/rank_service.go:
// 264-289), SortService.Sort runs once per request while requests overlap (sort/sort.go:65-125).  Every such call used
// to be one network round trip; here it would be one table pass (1.1 ms for one query, 1.4 for 32, 2.3 for 128, 3.1 for 256) or one
// launch-bound kernel chain.  The coalescer turns N concurrent single-request calls into one batch, for EVERY plug-in
// surface of a scene:
//
//   flavour      single-request call                         one batch =
//   recall       pg_coalescer_recall / _i2i_recall /         one table pass (the i2i trigger rows are gathered, the online
//                _online_recall                              user vectors run through the user tower, in front of it)
//   rank[a]      pg_coalescer_rank (DNN3 or FM + two-tower)  one rank launch per algorithm of RankAlgoList
//   recommend    pg_coalescer_recommend[_ex]                 recall → every rank algorithm → RankScore → sort → (DPPSort)
//   dpp          pg_coalescer_dpp                            KernelMatrix + DPPWithWindow for all requests of equal shape
//
//   caller threads ──push──► per-flavour queues ──► dispatcher thread ──► stream ──► completer thread ──► callers
//                                                  (forms a batch, copies the inputs                (waits for the batch's event,
//                                                   to pinned memory, enqueues the                   verifies the recall plan, wakes
//                                                   whole batch, never waits for the GPU)            the batch's callers: one futex)
//
// * A batch closes when it is full, or when nothing is in flight on the device and its oldest request has waited
//   max_wait_us.  While the device is busy an open batch just keeps growing — dispatching it early could not start
//   it any sooner — so under load batches fill up by themselves and an idle service answers within max_wait_us.
// * `depth` slots (pinned staging + device buffers + a PipeRun each) bound the batches in flight; a full batch is
//   enqueued behind the running one, so the stream never drains between batches.
// * Callers sleep in a futex wait on their request's state word (a cgo caller parks its OS thread, nothing spins);
//   every caller copies its own slice out of the slot's pinned output.
// * Deadlines (timeout_us; algorithm/eas/client.go:53-58 gives every predict 100 ms): a caller whose deadline passes
//   leaves with PG_ERR_TIMEOUT — out of the queue if its request was still waiting there, otherwise it abandons the
//   request record, which the workers retire when the batch completes.  The inputs of a staged request were copied,
//   outputs are only ever written by the caller itself, so nothing touches the caller's memory after it returned.
#include "pipeline.hpp"

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <deque>
#include <string>
#include <thread>

namespace pg {
namespace {

using Clock = std::chrono::steady_clock;

enum Flavour { kRecall = 0, kRank = 1, kRecommend = 2, kDpp = 3, kSsd = 4 };      // statistics index
enum Queue { kQRecall = 0, kQRecommend = 1, kQDpp = 2, kQRecallL2 = 3, kQRank0 = 4 };     // kQRank0 + algorithm index
constexpr uint32_t kL2Batch = 32;      // squared-Euclidean recalls per pass: the exact scan serves 32 queries at the price of one;
                                       // 128 when the table has an int8 shadow (the screened pass)
inline uint32_t l2_batch_limit(const pg_coalescer* c);
constexpr int kNumQueues = kQRank0 + kMaxAlgos;
enum QueryKind : uint32_t { kVector = 0, kTrigger = 1, kOnline = 2 };
enum ReqState : uint32_t { kQueued = 0, kStaging = 1, kStaged = 2, kDone = 3, kAbandoned = 4 };

struct Slot;

// the shape of a diversity re-rank call (DPPSort or SSDSort): calls of equal shape share a batched launch
struct DppKey {
    uint32_t n = 0, topn = 0, window = 0, hook_dim = 0;
    int normalize = 0, ensure_pos = 0, has_table = 0;
    int ssd = 0, star = 0;           // SSDSort: alpha carries gamma, star = UseSSDStar
    double alpha = 0.0;
    bool operator==(const DppKey& o) const {
        return n == o.n && topn == o.topn && window == o.window && hook_dim == o.hook_dim && normalize == o.normalize &&
               ensure_pos == o.ensure_pos && has_table == o.has_table && ssd == o.ssd && star == o.star && alpha == o.alpha;
    }
};

// One call.  Heap-allocated and reference-counted (the caller and the workers hold one reference each): a caller
// that gives up at its deadline must be able to leave while a worker still points at the record.
struct Req {
    int queue = 0;
    const float* vec = nullptr;            // query / user vector
    const int32_t* ufids = nullptr;        // FM + two-tower: the user's field ids
    uint32_t qkind = kVector;              // recall: what `vec` / `trigger_row` is
    uint32_t trigger_row = 0;
    const uint32_t* cand = nullptr;        // rank / dpp: candidate rows
    uint32_t n = 0;                        // rank / dpp: candidates; recommend: top_n
    const double* hook = nullptr;          // dpp: hook embeddings
    std::vector<double> rel;               // dpp: relevance scores as KernelMatrix uses them (normalised by the caller's thread)
    DppKey key;
    Clock::time_point arrived;
    // filled by the workers
    Slot* slot = nullptr;
    uint32_t index = 0;                    // position in the batch
    uint32_t item0 = 0;                    // rank: offset of the request's candidates in the batch
    int rc = PG_OK;
    char err[256] = {0};
    std::atomic<uint32_t> state{kQueued};  // futex word
    std::atomic<int> refs{2};
};

void req_unref(Req* r) {
    if (r->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete r;
}

struct Slot {
    int id = 0;
    int queue = 0;                         // which queue the current batch came from
    std::vector<Req*> reqs;
    uint32_t n_req = 0;                    // requests in the batch (reqs is handed back to the callers at wake-up)
    uint32_t n_items = 0;                  // rank / dpp: candidates in the batch; recommend: page width of the output image
    DppKey key;                            // dpp: the batch's shape
    PipeRun* run = nullptr;
    pg_ctx* ctx = nullptr;                 // the context (stream + scratch) this slot's batches run on
    hipEvent_t done = nullptr;             // behind the batch's device → host copies
    hipEvent_t computed = nullptr;         // behind its last kernel (the copy stream waits for it)
    Clock::time_point enqueued;
    std::atomic<uint32_t> pending{0};      // callers that have not copied their slice yet
    bool verified = false;                 // the recall plan of the current batch has been checked (a patched batch is only copied again)
    // pinned host staging
    float* h_vec = nullptr;                // [rows][vec_w]
    int32_t* h_ufid = nullptr;             // [rows][ufid_stride]
    uint32_t* h_qk = nullptr;              // recall: [max_batch] {kind, trigger row} pairs
    float* h_uq = nullptr;                 // recall: [max_batch][d_user of the query model]
    uint32_t* h_cand = nullptr;            // rank: concatenated candidate rows
    uint32_t* h_off = nullptr;             // rank: request offsets
    char* h_out = nullptr;                 // flavour-specific output image
    // device
    float* d_vec = nullptr;
    int32_t* d_ufid = nullptr;
    uint32_t* d_qk = nullptr;
    float* d_uq = nullptr;
    float* d_qemb = nullptr;               // recall: the query model's embeddings [max_batch][dim]
    uint32_t* d_cand = nullptr;
    uint32_t* d_off = nullptr;
    uint64_t* d_rows = nullptr;            // [max_batch][k]
    float* d_recall = nullptr;
    float* d_rank = nullptr;               // n_algos planes of rank_stride floats
    double* d_fused = nullptr;
    uint32_t* d_order = nullptr;
    uint32_t* d_count = nullptr;           // [max_batch]
    uint32_t* d_pick = nullptr;            // re-rank: [max_batch][max_top_n]
    uint32_t* d_pick_cnt = nullptr;
    char* d_page = nullptr;                // recommend: the pages, layout as h_out
    RecommendCall call;                    // recommend: what was enqueued (the verification may re-run single requests)
    // a coalescer over a shard group: the step's ticket and its pages (host memory, [max_batch][max_top_n] planes)
    pg_group_ticket* gticket = nullptr;
    uint64_t* g_rows = nullptr;
    float* g_rec = nullptr;
    float* g_rnk = nullptr;
    double* g_fus = nullptr;
    uint32_t* g_cnt = nullptr;
    // dpp flavour (allocated by the first such batch)
    bool dpp_ready = false;
    uint32_t* h_dcand = nullptr;
    double* h_drel = nullptr;
    double* h_dhook = nullptr;
    uint32_t* d_dcand = nullptr;
    double* d_drel = nullptr;
    double* d_dhook = nullptr;
    float* d_demb = nullptr;
    uint32_t* d_dout = nullptr;            // [items] picks, [max dpp batch] counts behind them
    uint32_t* h_dout = nullptr;
};

inline void futex_wait(std::atomic<uint32_t>* w, uint32_t expect, const timespec* rel_timeout) {
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expect, rel_timeout, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t>* w) {
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

// recall batches carry three kinds of query: the request's own vector (already in d_vec), a trigger row of the
// trigger table (I2IVectorRecall), or the query model's embedding of the user's features (OnlineVectorRecall)
__global__ void query_fixup_kernel(float* __restrict__ vec, const uint32_t* __restrict__ qk, const float* __restrict__ trig_tab,
                                   const float* __restrict__ qemb, uint32_t nq, uint32_t dim) {
    const uint32_t q = blockIdx.x;
    const uint32_t kind = qk[2 * q];
    if (kind == kVector) return;
    const float* src = kind == kTrigger ? trig_tab + (size_t)qk[2 * q + 1] * dim : qemb + (size_t)q * dim;
    for (uint32_t i = threadIdx.x; i < dim; i += blockDim.x) vec[(size_t)q * dim + i] = src[i];
}

__global__ void stall_kernel(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace
}  // namespace pg

struct pg_coalescer {
    pg_ctx* ctx = nullptr;
    bool timers = false;             // PG_COALESCER_TIMERS=1: the batches record the contexts' stage timers (pg_stats' last_*_ms) — ~50 us per small batch
    pg_ctx* sibling = nullptr;       // a second context on the same device (own stream and scratch): the slots alternate
                                     // between the two, so the latency-bound head and tail of one batch (pilot, selects,
                                     // fusion, sort) run under the other batch's scan / rank kernels
    const pg_table* t = nullptr;
    const pg_table* trigger_table = nullptr;
    const pg_model* query_model = nullptr;
    pg::RankAlgoRef algos[pg::kMaxAlgos];
    std::string algo_names[pg::kMaxAlgos];
    int n_algos = 0;
    int plane0[pg::kMaxAlgos] = {0, 1, 2, 3};  // first score plane of algorithm a (a multi-output DNN3 takes one per head)
    int n_planes = 0;
    uint32_t max_heads = 1;                // most outputs of any one algorithm (a rank-flavour batch writes that many planes)
    std::vector<std::string> plane_names;
    const pg_expr* e = nullptr;
    pg::ExprHold e_hold;             // (pg_expr_set_score_rewrites refuses while a coalescer holds the expression's bindings)
    std::vector<int> var_src;
    pg::RerankStage rerank;
    pg_group* group = nullptr;       // a coalescer over a shard group (pg_coalescer_create_group): recommend only
    pg_group_plan gplan{};
    std::string grank_var;
    std::atomic<uint32_t> outstanding{0};            // requests between submit and return (the router's load measure)
    uint32_t k = 0, max_batch = 0, max_wait_us = 0, depth = 0, max_top_n = 0, max_rank_items = 0, timeout_us = 0;
    // arrival statistics per queue (under mu): a request that arrives at an idle device only waits for company when company is likely —
    // when the recent inter-arrival gap (EWMA) is below max_wait_us; a lone caller is dispatched at once
    pg::Clock::time_point last_arrival[pg::kNumQueues];
    double gap_ewma_us[pg::kNumQueues] = {};
    bool seen_arrival[pg::kNumQueues] = {};
    // Rejoin hold (recall-based queues, under mu).  Callers are closed loops (a goroutine blocked in Recall.GetCandidateItems /
    // IAlgorithm.Run issues its next request when this one returns), so the n callers of a batch that just completed are back
    // within some tens of microseconds — and a partial batch dispatched the moment the device falls idle splits them from the
    // requests that waited meanwhile into two cohorts that alternate for ever (32 callers: two batches of ~16, every request
    // waits for the other cohort's pass: p50 3.2 ms where one batch of 32 takes 1.8).  After a completion the waiting partial
    // batch is therefore held until those n are back, at most rejoin window (50 + 2 n us); rejoin_score tracks how many came
    // back in time (open-loop arrivals do not): below one half the hold is off, re-tried every sixteenth completion.
    pg::Clock::time_point rejoin_until[pg::kNumQueues];
    size_t rejoin_target[pg::kNumQueues] = {};
    uint32_t rejoin_n[pg::kNumQueues] = {};
    double rejoin_score[pg::kNumQueues] = {};
    uint32_t rejoin_probe[pg::kNumQueues] = {};
    uint32_t max_rank_reqs = 0, rank_item_cap = 0;
    uint32_t dim = 0, vec_w = 0, ufid_stride = 0, vec_rows = 0;
    uint32_t dpp_item_cap = 0, dpp_max_n = 0, dpp_max_hook = 0;
    size_t rank_stride = 0;
    hipStream_t copy_stream = nullptr;

    std::mutex mu;                                   // queues, slots, stop
    std::condition_variable cv_dispatch;             // new request, slot freed, batch completed
    std::condition_variable cv_complete;             // batch enqueued
    std::deque<pg::Req*> queue[pg::kNumQueues];
    uint64_t queue_items[pg::kNumQueues] = {0};      // rank queues: candidates waiting (kept in step with the deque under mu)
    std::vector<pg::Slot*> slots;
    std::vector<pg::Slot*> free_slots;
    std::deque<pg::Slot*> inflight;
    bool stop = false;
    bool broken = false;                             // a device error surfaced: every call fails with PG_ERR_DEVICE from now on
    std::string broken_msg;
    std::thread dispatcher, completer;
    pg_coalescer_stats_t stats{};
};

namespace pg {
namespace {

inline uint32_t l2_batch_limit(const pg_coalescer* c) {
    const bool screened = c->t->dim == 128 && c->t->stats_valid && c->t->all_finite && c->t->shadow_is_i8 && !c->ctx->knobs.l2_exact;
    return std::min(c->max_batch, screened ? 128u : kL2Batch);
}
int flavour_of(int queue) {
    return (queue == kQRecall || queue == kQRecallL2) ? kRecall : (queue == kQRecommend ? kRecommend : (queue == kQDpp ? kDpp : kRank));
}

size_t page_bytes(const pg_coalescer* c) { return (size_t)c->max_batch * c->max_top_n * page_entry_bytes(std::max(c->n_planes, 1)); }

void fail_req(Req* r, int rc, const char* msg) {
    r->rc = rc;
    snprintf(r->err, sizeof r->err, "%s", msg);
}

// how many DPP requests of `n` candidates one batch takes: bounded by the staging buffers and by the kernel
// matrices (n^2 doubles per request in the context's scratch: at most 1 GiB of them per batch)
uint32_t dpp_batch_limit(const pg_coalescer* c, uint32_t n) {
    uint64_t b = c->dpp_item_cap / std::max(n, 1u);
    b = std::min<uint64_t>(b, (1ull << 27) / ((uint64_t)n * n));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(b, kMaxQueries));
}

int alloc_slot(pg_coalescer* c, Slot* s) {
    if (c->group) {                      // host staging only: the group owns every device buffer
        const size_t np = (size_t)c->max_batch * c->max_top_n;
        PG_HIP(hipHostMalloc((void**)&s->h_vec, (size_t)c->max_batch * c->dim * 4));
        s->g_rows = new uint64_t[np];
        s->g_rec = new float[np];
        s->g_rnk = new float[np];
        s->g_fus = new double[np];
        s->g_cnt = new uint32_t[c->max_batch];
        return PG_OK;
    }
    const size_t nb = c->max_batch, k = c->k;
    const bool rank = c->n_algos > 0;
    PG_HIP(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
    PG_HIP(hipEventCreateWithFlags(&s->computed, hipEventDisableTiming));
    PG_HIP(hipHostMalloc((void**)&s->h_vec, (size_t)c->vec_rows * c->vec_w * 4));
    PG_HIP(hipMalloc((void**)&s->d_vec, (size_t)c->vec_rows * c->vec_w * 4));
    PG_HIP(hipHostMalloc((void**)&s->h_qk, nb * 8));
    PG_HIP(hipMalloc((void**)&s->d_qk, nb * 8));
    if (c->ufid_stride) {
        PG_HIP(hipHostMalloc((void**)&s->h_ufid, (size_t)c->vec_rows * c->ufid_stride * 4));
        PG_HIP(hipMalloc((void**)&s->d_ufid, (size_t)c->vec_rows * c->ufid_stride * 4));
    }
    if (c->query_model) {
        PG_HIP(hipHostMalloc((void**)&s->h_uq, nb * c->query_model->d_user * 4));
        PG_HIP(hipMalloc((void**)&s->d_uq, nb * c->query_model->d_user * 4));
        PG_HIP(hipMalloc((void**)&s->d_qemb, nb * c->dim * 4));
    }
    size_t out_bytes = nb * k * 12 + nb * 4;                       // recall image: rows | scores | counts
    if (rank) {
        PG_HIP(hipHostMalloc((void**)&s->h_cand, (size_t)c->rank_item_cap * 4));
        PG_HIP(hipHostMalloc((void**)&s->h_off, ((size_t)c->max_rank_reqs + 1) * 4));
        PG_HIP(hipMalloc((void**)&s->d_cand, (size_t)c->rank_item_cap * 4));
        PG_HIP(hipMalloc((void**)&s->d_off, ((size_t)c->max_rank_reqs + 1) * 4));
        out_bytes = std::max(out_bytes, (size_t)c->rank_item_cap * 4 * c->max_heads);    // (a multi-output model: a plane per head)
    }
    if (c->e) out_bytes = std::max(out_bytes, page_bytes(c) + nb * 8);
    PG_HIP(hipHostMalloc((void**)&s->h_out, out_bytes));
    PG_HIP(hipMalloc((void**)&s->d_rows, nb * k * 8));
    PG_HIP(hipMalloc((void**)&s->d_recall, nb * k * 4));
    PG_HIP(hipMalloc((void**)&s->d_count, nb * 4));
    if (rank) PG_HIP(hipMalloc((void**)&s->d_rank, (size_t)std::max(c->n_planes, (int)c->max_heads) * c->rank_stride * 4));
    if (c->e) {
        PG_HIP(hipMalloc((void**)&s->d_fused, nb * k * 8));
        PG_HIP(hipMalloc((void**)&s->d_order, nb * k * 4));
        PG_HIP(hipMalloc((void**)&s->d_page, page_bytes(c)));
        if (c->rerank.kind) {
            PG_HIP(hipMalloc((void**)&s->d_pick, nb * c->max_top_n * 4));
            PG_HIP(hipMalloc((void**)&s->d_pick_cnt, nb * 4));
        }
    }
    return pipe_run_acquire(s->ctx, &s->run);
}

// the DPP flavour's buffers: allocated by the first batch that needs them (dispatcher thread)
int ensure_dpp_buffers(pg_coalescer* c, Slot* s) {
    if (s->dpp_ready) return PG_OK;
    PG_HIP(hipSetDevice(s->ctx->device));
    const size_t items = c->dpp_item_cap;
    PG_HIP(hipHostMalloc((void**)&s->h_dcand, items * 4));
    PG_HIP(hipHostMalloc((void**)&s->h_drel, items * 8));
    PG_HIP(hipMalloc((void**)&s->d_dcand, items * 4));
    PG_HIP(hipMalloc((void**)&s->d_drel, items * 8));
    PG_HIP(hipMalloc((void**)&s->d_demb, items * c->dim * 4));
    PG_HIP(hipMalloc((void**)&s->d_dout, (items + kMaxQueries) * 4));
    PG_HIP(hipHostMalloc((void**)&s->h_dout, (items + kMaxQueries) * 4));
    if (c->dpp_max_hook) {
        PG_HIP(hipHostMalloc((void**)&s->h_dhook, items * c->dpp_max_hook * 8));
        PG_HIP(hipMalloc((void**)&s->d_dhook, items * c->dpp_max_hook * 8));
    }
    s->dpp_ready = true;
    return PG_OK;
}

void free_slot(pg_coalescer* c, Slot* s) {
    delete[] s->g_rows;
    delete[] s->g_rec;
    delete[] s->g_rnk;
    delete[] s->g_fus;
    delete[] s->g_cnt;
    if (s->run) pipe_run_release(s->ctx, s->run);
    if (s->done) hipEventDestroy(s->done);
    if (s->computed) hipEventDestroy(s->computed);
    for (void* p : {(void*)s->h_vec, (void*)s->h_ufid, (void*)s->h_qk, (void*)s->h_uq, (void*)s->h_cand, (void*)s->h_off, (void*)s->h_out,
                    (void*)s->h_dcand, (void*)s->h_drel, (void*)s->h_dhook, (void*)s->h_dout})
        if (p) hipHostFree(p);
    for (void* p : {(void*)s->d_vec, (void*)s->d_ufid, (void*)s->d_qk, (void*)s->d_uq, (void*)s->d_qemb, (void*)s->d_cand, (void*)s->d_off,
                    (void*)s->d_rows, (void*)s->d_recall, (void*)s->d_rank, (void*)s->d_fused, (void*)s->d_order, (void*)s->d_count,
                    (void*)s->d_pick, (void*)s->d_pick_cnt, (void*)s->d_page, (void*)s->d_dcand, (void*)s->d_drel, (void*)s->d_dhook,
                    (void*)s->d_demb, (void*)s->d_dout})
        if (p) hipFree(p);
    delete s;
}

// Outputs device → pinned host on the copy stream (so the next batch's kernels do not queue behind a PCIe transfer),
// then the completion event.  Called behind the batch's kernels, and again when the verification patched single
// requests in place.
int slot_copy_out(pg_coalescer* c, Slot* s) {
    hipStream_t st = s->ctx->stream;
    const uint32_t nq = s->n_req;
    const int fl = flavour_of(s->queue);
    if (fl == kRecommend) {
        // the largest page any request of the batch asked for
        const uint32_t top = s->n_items;
        int rc;
        if ((rc = page_launch(st, s->d_order, c->rerank.kind ? s->d_pick : nullptr, s->d_pick_cnt, s->d_rows, s->d_recall, s->d_rank,
                              c->rank_stride, c->n_planes, s->d_fused, nq, c->k, top, s->d_page)))
            return rc;
    }
    PG_HIP(hipEventRecord(s->computed, st));
    PG_HIP(hipStreamWaitEvent(c->copy_stream, s->computed, 0));
    if (fl == kRank) {
        // head o of a multi-output model: plane o of the batch, [n_items] each, back to back in the image
        const uint32_t heads = c->algos[s->queue - kQRank0].m->n_out;
        for (uint32_t o = 0; o < heads; ++o)
            PG_HIP(hipMemcpyAsync(s->h_out + (size_t)o * s->n_items * 4, s->d_rank + (size_t)o * c->rank_stride, (size_t)s->n_items * 4,
                                  hipMemcpyDeviceToHost, c->copy_stream));
    } else if (fl == kRecall) {
        const size_t nk = (size_t)nq * c->k;
        PG_HIP(hipMemcpyAsync(s->h_out, s->d_rows, nk * 8, hipMemcpyDeviceToHost, c->copy_stream));
        PG_HIP(hipMemcpyAsync(s->h_out + (size_t)c->max_batch * c->k * 8, s->d_recall, nk * 4, hipMemcpyDeviceToHost, c->copy_stream));
    } else if (fl == kDpp) {
        const size_t np = (size_t)nq * s->key.topn;
        PG_HIP(hipMemcpyAsync(s->h_dout, s->d_dout, np * 4, hipMemcpyDeviceToHost, c->copy_stream));
        PG_HIP(hipMemcpyAsync(s->h_dout + c->dpp_item_cap, s->d_dout + c->dpp_item_cap, (size_t)nq * 4, hipMemcpyDeviceToHost, c->copy_stream));
    } else {
        const size_t np = (size_t)nq * s->n_items;
        PG_HIP(hipMemcpyAsync(s->h_out, s->d_page, np * page_entry_bytes(c->n_planes), hipMemcpyDeviceToHost, c->copy_stream));
        PG_HIP(hipMemcpyAsync(s->h_out + page_bytes(c), s->d_count, (size_t)nq * 4, hipMemcpyDeviceToHost, c->copy_stream));
        if (c->rerank.kind)
            PG_HIP(hipMemcpyAsync(s->h_out + page_bytes(c) + (size_t)c->max_batch * 4, s->d_pick_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost,
                                  c->copy_stream));
    }
    PG_HIP(hipEventRecord(s->done, c->copy_stream));
    return PG_OK;
}

// the recall job of a recall-flavour batch, with its queries put together in front of it
int enqueue_recall_batch(pg_coalescer* c, Slot* s, bool first) {
    pg_ctx* ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t nq = s->n_req;
    int rc;
    std::lock_guard<std::mutex> g(ctx->mu);
    TimersScope quiet(ctx, c->timers);             // (nobody reads the stage timers of a coalesced batch: no event records)
    TableRead2 tr(c->t, c->trigger_table);
    RecallJob& j = s->run->job;
    // a swap / upload between this batch's first pass and its re-plan: the job's statistics and plans are the old version's —
    // the whole batch starts over on the new rows (one version per batch)
    if (!first && j.table_gen != c->t->generation.load(std::memory_order_relaxed)) first = true;
    if (first) {
        bool any_trigger = false, any_online = false;
        for (uint32_t q = 0; q < nq; ++q) {
            any_trigger = any_trigger || s->h_qk[2 * q] == kTrigger;
            any_online = any_online || s->h_qk[2 * q] == kOnline;
        }
        if (any_trigger || any_online) {
            PG_HIP(hipMemcpyAsync(s->d_qk, s->h_qk, (size_t)nq * 8, hipMemcpyHostToDevice, st));
            if (any_online) {
                PG_HIP(hipMemcpyAsync(s->d_uq, s->h_uq, (size_t)nq * c->query_model->d_user * 4, hipMemcpyHostToDevice, st));
                if ((rc = fm2t_user_embedding_locked(ctx, c->query_model, s->d_uq, nq, s->d_qemb))) return rc;
            }
            query_fixup_kernel<<<nq, 64, 0, st>>>(s->d_vec, s->d_qk, c->trigger_table->d, s->d_qemb, nq, c->dim);
            PG_HIP(hipGetLastError());
        }
        j = RecallJob();
        j.ctx = ctx;
        j.t = c->t;
        j.d_queries = s->d_vec;
        j.nq = nq;
        j.k = c->k;
        j.d_out_rows = s->d_rows;
        j.d_out_scores = s->d_recall;
        j.d_out_count = nullptr;
        j.h_status = s->run->h_status;
        j.events = &s->run->events;
        j.l2 = s->queue == kQRecallL2;
        if ((rc = recall_job_prepare(&j))) return rc;
    }
    s->run->patched = false;
    return recall_job_enqueue(&j);
}

int enqueue_rank_batch(pg_coalescer* c, Slot* s) {
    pg_ctx* ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t nq = s->n_req;
    const RankAlgoRef& al = c->algos[s->queue - kQRank0];
    const uint32_t du = al.m->d_user;
    PG_HIP(hipMemcpyAsync(s->d_vec, s->h_vec, (size_t)nq * du * 4, hipMemcpyHostToDevice, st));
    PG_HIP(hipMemcpyAsync(s->d_cand, s->h_cand, (size_t)s->n_items * 4, hipMemcpyHostToDevice, st));
    PG_HIP(hipMemcpyAsync(s->d_off, s->h_off, ((size_t)nq + 1) * 4, hipMemcpyHostToDevice, st));
    std::lock_guard<std::mutex> g(ctx->mu);
    TimersScope quiet(ctx, c->timers);             // (nobody reads the stage timers of a coalesced batch: no event records)
    TableRead tr(c->t->rw);
    if (al.m->kind != PG_MODEL_DNN3) PG_HIP(hipMemcpyAsync(s->d_ufid, s->h_ufid, (size_t)nq * al.m->nuf * 4, hipMemcpyHostToDevice, st));
    return rank_algo_locked(ctx, al, c->t, s->d_vec, s->d_ufid, s->d_cand, s->d_off, nq, s->n_items, s->d_rank, c->rank_stride);
}

int enqueue_dpp_batch(pg_coalescer* c, Slot* s) {
    pg_ctx* ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t R = s->n_req;
    const DppKey& key = s->key;
    const size_t items = (size_t)R * key.n;
    PG_HIP(hipMemcpyAsync(s->d_drel, s->h_drel, items * 8, hipMemcpyHostToDevice, st));
    if (key.has_table) PG_HIP(hipMemcpyAsync(s->d_dcand, s->h_dcand, items * 4, hipMemcpyHostToDevice, st));
    if (key.hook_dim) PG_HIP(hipMemcpyAsync(s->d_dhook, s->h_dhook, items * key.hook_dim * 8, hipMemcpyHostToDevice, st));
    std::lock_guard<std::mutex> g(ctx->mu);
    TimersScope quiet(ctx, c->timers);             // (nobody reads the stage timers of a coalesced batch: no event records)
    TableRead tr(c->t->rw);
    int rc;
    if (key.ssd)
        return ssd_run_locked(ctx, c->t, s->d_dcand, s->d_drel, R, key.n, key.alpha, key.topn, key.window, key.normalize, key.ensure_pos,
                              key.star, s->d_dout);
    if (key.has_table && (rc = table_gather_locked(ctx, c->t, s->d_dcand, (uint32_t)items, s->d_demb))) return rc;
    return dpp_run_locked(ctx, key.has_table ? s->d_demb : nullptr, key.hook_dim ? s->d_dhook : nullptr, s->d_drel, R, key.n,
                          key.has_table ? c->dim : 0u, key.hook_dim, key.alpha, key.topn, key.window, key.normalize, key.ensure_pos,
                          key.has_table, s->d_dout, s->d_dout + c->dpp_item_cap);
}

// Enqueue the slot's batch: inputs host → device, the kernels, the copy-out.  first = false: the recall plan of a
// recall / recommend batch did not hold; run its next plan and everything behind it again.
int slot_enqueue(pg_coalescer* c, Slot* s, bool first) {
    if (c->group) {
        // one step of the shard group for the whole batch: enqueued on every shard, collected by the completer
        uint32_t top = 1;
        for (const Req* r : s->reqs) top = std::max(top, r->n);
        s->n_items = top;
        return pg_group_recommend_begin(c->group, c->e, c->grank_var.c_str(), &c->gplan, s->h_vec, s->n_req, top, &s->gticket);
    }
    pg_ctx* ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t nq = (uint32_t)s->n_req;
    const int fl = flavour_of(s->queue);
    int rc;
    if (fl == kRank) {
        if ((rc = enqueue_rank_batch(c, s))) return rc;
        return slot_copy_out(c, s);
    }
    if (fl == kDpp) {
        if ((rc = enqueue_dpp_batch(c, s))) return rc;
        return slot_copy_out(c, s);
    }
    if (first) PG_HIP(hipMemcpyAsync(s->d_vec, s->h_vec, (size_t)nq * c->dim * 4, hipMemcpyHostToDevice, st));
    if (fl == kRecall) {
        if ((rc = enqueue_recall_batch(c, s, first))) return rc;
        return slot_copy_out(c, s);
    }
    // recommend
    if (first && c->ufid_stride) PG_HIP(hipMemcpyAsync(s->d_ufid, s->h_ufid, (size_t)nq * c->ufid_stride * 4, hipMemcpyHostToDevice, st));
    uint32_t top = 1;
    for (const Req* r : s->reqs) top = std::max(top, r->n);
    s->n_items = top;
    RecommendCall& call = s->call;
    call = RecommendCall();
    call.t = c->t;
    for (int a = 0; a < c->n_algos; ++a) call.algos[a] = c->algos[a];
    call.n_algos = c->n_algos;
    for (int a = 0; a < c->n_algos; ++a) call.plane0[a] = c->plane0[a];
    call.n_planes = c->n_planes;
    call.e = c->e;
    call.var_src = c->var_src.data();
    call.nv = pg_expr_num_vars(c->e);
    call.d_queries = s->d_vec;
    call.d_ufids = s->d_ufid;
    call.ufid_stride = c->ufid_stride;
    call.nq = nq;
    call.k = c->k;
    call.d_rows = s->d_rows;
    call.d_recall = s->d_recall;
    call.d_rank = s->d_rank;
    call.rank_stride = c->rank_stride;
    call.d_fused = s->d_fused;
    call.d_order = s->d_order;
    call.d_count = s->d_count;
    call.pads = c->t->rows < c->k;
    call.rerank = c->rerank;
    call.top_n = top;
    call.d_pick = s->d_pick;
    call.d_pick_cnt = s->d_pick_cnt;
    call.timers = c->timers;
    if ((rc = recommend_enqueue(ctx, call, s->run, first))) return rc;
    return slot_copy_out(c, s);
}

// finish every request of a slot with (rc, message of this thread)
void slot_fail(Slot* s, int rc) {
    const char* msg = pg_last_error();
    for (Req* r : s->reqs) fail_req(r, rc, msg);
}

void release_slot(pg_coalescer* c, Slot* s) {
    if (s->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        {
            std::lock_guard<std::mutex> g(c->mu);
            c->free_slots.push_back(s);
        }
        c->cv_dispatch.notify_one();
    }
}

// hand a finished request to its caller — or retire it when the caller left at its deadline
void finish_req(pg_coalescer* c, Req* r, Slot* s) {
    const uint32_t prev = r->state.exchange(kDone, std::memory_order_acq_rel);
    if (prev == kAbandoned) {
        if (s) release_slot(c, s);
    } else {
        futex_wake_all(&r->state);
    }
    req_unref(r);
}

void slot_wake(pg_coalescer* c, Slot* s) {
    // callers release the slot: the last one to have copied its slice returns it to the free list
    s->pending.store((uint32_t)s->reqs.size(), std::memory_order_release);
    std::vector<Req*> reqs;
    reqs.swap(s->reqs);                    // a woken caller may free its record at once: do not touch it after finish_req
    for (Req* r : reqs) finish_req(c, r, s);
}

// the batch taken from queue `kind` into slot `s` (caller holds c->mu)
void take_batch(pg_coalescer* c, int kind, Slot* s) {
    std::deque<Req*>& q = c->queue[kind];
    s->queue = kind;
    s->reqs.clear();
    s->n_items = 0;
    s->verified = false;
    if (kind >= kQRank0) {
        while (!q.empty() && s->reqs.size() < c->max_rank_reqs && s->n_items + q.front()->n <= c->rank_item_cap) {
            Req* r = q.front();
            q.pop_front();
            r->item0 = s->n_items;
            s->n_items += r->n;
            c->queue_items[kind] -= r->n;
            s->reqs.push_back(r);
        }
    } else if (kind == kQDpp) {
        // every request of the head's shape, up to what one batch holds
        s->key = q.front()->key;
        const uint32_t limit = dpp_batch_limit(c, s->key.n);
        for (auto it = q.begin(); it != q.end() && s->reqs.size() < limit;) {
            if ((*it)->key == s->key) {
                s->reqs.push_back(*it);
                it = q.erase(it);
            } else {
                ++it;
            }
        }
        s->n_items = (uint32_t)s->reqs.size() * s->key.n;
    } else {
        const uint32_t limit = kind == kQRecallL2 ? l2_batch_limit(c) : c->max_batch;
        while (!q.empty() && s->reqs.size() < limit) {
            s->reqs.push_back(q.front());
            q.pop_front();
        }
    }
    for (Req* r : s->reqs) r->state.store(kStaging, std::memory_order_release);
}

// copy the callers' inputs into the slot's pinned staging (the callers are blocked, or spinning until kStaged at
// their deadline: their buffers are stable)
void stage_batch(pg_coalescer* c, Slot* s) {
    const int kind = s->queue;
    const uint32_t nq = (uint32_t)s->reqs.size();
    s->n_req = nq;
    if (kind == kQDpp) {
        const DppKey& key = s->key;
        for (uint32_t i = 0; i < nq; ++i) {
            Req* r = s->reqs[i];
            r->slot = s;
            r->index = i;
            memcpy(s->h_drel + (size_t)i * key.n, r->rel.data(), (size_t)key.n * 8);
            if (key.has_table) memcpy(s->h_dcand + (size_t)i * key.n, r->cand, (size_t)key.n * 4);
            if (key.hook_dim) memcpy(s->h_dhook + (size_t)i * key.n * key.hook_dim, r->hook, (size_t)key.n * key.hook_dim * 8);
        }
    } else {
        const RankAlgoRef* al = kind >= kQRank0 ? &c->algos[kind - kQRank0] : nullptr;
        const uint32_t w = al ? al->m->d_user : c->dim;
        const uint32_t uw = al ? al->m->nuf : c->ufid_stride;
        for (uint32_t i = 0; i < nq; ++i) {
            Req* r = s->reqs[i];
            r->slot = s;
            r->index = i;
            if (kind == kQRecall) {
                s->h_qk[2 * i] = r->qkind;
                s->h_qk[2 * i + 1] = r->trigger_row;
                if (r->qkind == kVector) memcpy(s->h_vec + (size_t)i * w, r->vec, (size_t)w * 4);
                else memset(s->h_vec + (size_t)i * w, 0, (size_t)w * 4);
                if (r->qkind == kOnline) memcpy(s->h_uq + (size_t)i * c->query_model->d_user, r->vec, (size_t)c->query_model->d_user * 4);
                else if (c->query_model) memset(s->h_uq + (size_t)i * c->query_model->d_user, 0, (size_t)c->query_model->d_user * 4);
            } else {
                memcpy(s->h_vec + (size_t)i * w, r->vec, (size_t)w * 4);
                if (kind == kQRecallL2) {              // plain vector queries (enqueue_recall_batch looks at the kinds)
                    s->h_qk[2 * i] = kVector;
                    s->h_qk[2 * i + 1] = 0;
                }
            }
            if (uw && kind != kQRecall && kind != kQRecallL2) {
                if (r->ufids) memcpy(s->h_ufid + (size_t)i * uw, r->ufids, (size_t)uw * 4);
                else memset(s->h_ufid + (size_t)i * uw, 0, (size_t)uw * 4);
            }
            if (al) {
                s->h_off[i] = r->item0;
                memcpy(s->h_cand + r->item0, r->cand, (size_t)r->n * 4);
            }
        }
        if (al) s->h_off[nq] = s->n_items;
    }
    for (Req* r : s->reqs) {
        uint32_t expect = kStaging;
        r->state.compare_exchange_strong(expect, kStaged, std::memory_order_acq_rel);
    }
}

void dispatcher_main(pg_coalescer* c) {
    hipSetDevice(c->ctx->device);
    std::unique_lock<std::mutex> lk(c->mu);
    while (true) {
        if (c->stop) break;
        // Which queue goes next?  A batch is ready when it is full, or when nothing is in flight on the device and
        // its oldest request has waited max_wait_us; among ready queues the one whose head is oldest wins.
        const bool idle = c->inflight.empty();
        const auto now = Clock::now();
        int kind = -1;
        bool any = false;
        auto earliest = Clock::time_point::max();
        for (int f = 0; f < kNumQueues; ++f) {
            std::deque<Req*>& qf = c->queue[f];
            if (qf.empty()) continue;
            any = true;
            bool full;
            if (f >= kQRank0) {
                full = qf.size() >= c->max_rank_reqs || c->queue_items[f] >= (uint64_t)c->max_batch * c->k;
            } else if (f == kQDpp) {
                full = qf.size() >= dpp_batch_limit(c, qf.front()->key.n);
            } else {
                full = qf.size() >= (f == kQRecallL2 ? l2_batch_limit(c) : c->max_batch);
            }
            const uint32_t wait_us = c->gap_ewma_us[f] > (double)c->max_wait_us ? 0u : c->max_wait_us;
            auto deadline = qf.front()->arrived + std::chrono::microseconds(wait_us);
            // (rejoin hold: everyone the last batch answered is back -> go at once; else not before the window closes)
            const bool holding = now < c->rejoin_until[f];
            const bool rejoined = holding && qf.size() >= c->rejoin_target[f];
            if (rejoined) deadline = now;
            else if (holding && c->rejoin_until[f] > deadline) deadline = c->rejoin_until[f];
            // recall-based flavours: a table pass costs the same for 1 query as for 256, so a partial batch only goes out
            // when the device has nothing to do; rank / DPP launches cost what their items cost, so a partial batch goes out
            // as soon as its head has waited (a free slot permitting) and pipelines behind the running one
            const bool per_item = f >= kQRank0 || f == kQDpp;
            if ((idle || per_item) && deadline < earliest) earliest = deadline;      // (a deadline that time alone will make ready)
            if ((full || ((idle || per_item) && now >= deadline)) && (kind < 0 || qf.front()->arrived < c->queue[kind].front()->arrived)) kind = f;
        }
        if (kind < 0 || c->free_slots.empty()) {
            // nothing ready (or `depth` batches already in flight: keep collecting).  A new request, a completion
            // or a released slot wakes us; an idle device additionally at the oldest request's deadline.
            if (any && kind < 0 && !c->free_slots.empty() && earliest != Clock::time_point::max()) c->cv_dispatch.wait_until(lk, earliest);
            else c->cv_dispatch.wait(lk);
            continue;
        }
        Slot* s = c->free_slots.front();               // oldest first: consecutive batches alternate between the contexts
        c->free_slots.erase(c->free_slots.begin());
        if (c->rejoin_n[kind]) {                       // how much of the last batch's callers made it into this one
            const size_t base = c->rejoin_target[kind] - c->rejoin_n[kind];
            const size_t have = c->queue[kind].size();
            const double frac = have <= base ? 0.0 : (have - base >= c->rejoin_n[kind] ? 1.0 : (double)(have - base) / c->rejoin_n[kind]);
            c->rejoin_score[kind] = 0.8 * c->rejoin_score[kind] + 0.2 * frac;
            c->rejoin_n[kind] = 0;
            c->rejoin_until[kind] = Clock::time_point();
        }
        take_batch(c, kind, s);
        lk.unlock();
        int rc = kind == kQDpp ? ensure_dpp_buffers(c, s) : PG_OK;
        if (!rc) {
            stage_batch(c, s);
            s->enqueued = Clock::now();
            rc = slot_enqueue(c, s, true);
        } else {
            s->n_req = (uint32_t)s->reqs.size();
            for (Req* r : s->reqs) {
                r->slot = s;
                uint32_t expect = kStaging;
                r->state.compare_exchange_strong(expect, kStaged, std::memory_order_acq_rel);
            }
        }
        lk.lock();
        const int fl = (kind == kQDpp && s->key.ssd) ? (int)kSsd : flavour_of(kind);
        c->stats.requests[fl] += s->n_req;
        c->stats.batches[fl] += 1;
        c->stats.largest_batch[fl] = std::max<uint64_t>(c->stats.largest_batch[fl], s->n_req);
        if (rc) {
            slot_fail(s, rc);
            lk.unlock();
            slot_wake(c, s);
            lk.lock();
            continue;
        }
        c->inflight.push_back(s);
        c->cv_complete.notify_one();
    }
    // shutdown: nothing new is dispatched; whatever still waits fails
    for (auto& q : c->queue)
        while (!q.empty()) {
            Req* r = q.front();
            q.pop_front();
            fail_req(r, PG_ERR_INVALID, "pg_coalescer: destroyed while the request was waiting");
            r->slot = nullptr;
            r->state.store(kStaged, std::memory_order_release);
            lk.unlock();
            finish_req(c, r, nullptr);
            lk.lock();
        }
}

void completer_main(pg_coalescer* c) {
    hipSetDevice(c->ctx->device);
    std::unique_lock<std::mutex> lk(c->mu);
    while (true) {
        if (c->inflight.empty()) {
            if (c->stop) break;
            c->cv_complete.wait(lk);
            continue;
        }
        Slot* s = c->inflight.front();
        lk.unlock();
        int rc = PG_OK;
        bool replanned = false, device_fault = false;
        const int fl = flavour_of(s->queue);
        for (; !c->group;) {
            if (hipEventSynchronize(s->done) != hipSuccess) {
                set_error("pg_coalescer: %s", hipGetErrorString(hipGetLastError()));
                rc = PG_ERR_DEVICE;
                device_fault = true;
                break;
            }
            if (fl == kRank || fl == kDpp || s->verified) break;
            bool ok = false;
            // (recall_job_check + finish under ctx->mu; a few failed requests are re-run in place)
            if ((rc = recommend_verify(s->ctx, s->run, &ok, fl == kRecommend ? &s->call : nullptr))) break;
            if (ok) s->verified = true;
            if (ok && s->run->patched) {                 // device outputs changed after the copy-out: copy again
                s->run->patched = false;
                replanned = true;
                if ((rc = slot_copy_out(c, s))) break;
                continue;
            }
            if (ok) break;
            replanned = true;
            if ((rc = slot_enqueue(c, s, false))) break;
        }
        if (c->group) {
            // waits for the step, verifies every shard's plan (re-running the step where one failed), assembles the pages
            rc = pg_group_recommend_end(c->group, s->gticket, s->g_rows, s->g_rec, s->g_rnk, s->g_fus, s->g_cnt);
            s->gticket = nullptr;
            device_fault = rc == PG_ERR_DEVICE;
        }
        if (rc) {
            slot_fail(s, rc);
        } else if (fl == kRecommend && !c->group) {
            for (Req* r : s->reqs)
                if (s->run->h_status[kExprFlagAt + r->index]) {
                    set_expr_arith_error(c->e);
                    fail_req(r, PG_ERR_ARITH, pg_last_error());
                }
        }
        const double ms = std::chrono::duration<double, std::milli>(Clock::now() - s->enqueued).count();
        lk.lock();
        c->inflight.pop_front();
        if (!c->group && s->queue < kQRank0 && s->queue != kQDpp && s->n_req && c->inflight.empty() &&
            c->ctx->knobs.coalescer_rejoin) {
            const int f = s->queue;
            const bool on = c->rejoin_score[f] >= 0.5 || (++c->rejoin_probe[f] & 15u) == 0;
            if (on) {
                c->rejoin_n[f] = s->n_req;
                c->rejoin_target[f] = c->queue[f].size() + s->n_req;
                c->rejoin_until[f] = Clock::now() + std::chrono::microseconds(50 + (int)(c->ctx->knobs.coalescer_rejoin_us_per_caller * s->n_req));
            }
        }
        c->stats.device_ms[(fl == kDpp && s->key.ssd) ? (int)kSsd : fl] += ms;
        if (replanned) c->stats.replans++;
        std::vector<Req*> orphans;
        if (device_fault) {
            // a HIP error is sticky: nothing queued behind it will run.  Fail what waits, refuse what comes; the host
            // creates a fresh context (or a fresh process) and a new coalescer.
            c->broken = true;
            c->broken_msg = pg_last_error();
            for (auto& q : c->queue) {
                for (Req* r : q) {
                    fail_req(r, PG_ERR_DEVICE, c->broken_msg.c_str());
                    r->slot = nullptr;
                    r->state.store(kStaged, std::memory_order_release);     // out of the queue: from here on a caller can only abandon it
                    orphans.push_back(r);
                }
                q.clear();
                c->queue_items[&q - c->queue] = 0;
            }
        }
        lk.unlock();
        for (Req* r : orphans) finish_req(c, r, nullptr);
        slot_wake(c, s);
        lk.lock();
        c->cv_dispatch.notify_one();               // the device may be idle now: a waiting partial batch can go
    }
}

// caller side: queue the request, sleep until a worker finished it (or the deadline passes).  Returns PG_OK when
// the request was finished (r->rc holds its status), PG_ERR_TIMEOUT / PG_ERR_* when the caller leaves without it —
// in that case the record must not be touched any more.
int submit_and_wait_inner(pg_coalescer* c, Req* r);
int submit_and_wait(pg_coalescer* c, Req* r) {
    c->outstanding.fetch_add(1, std::memory_order_relaxed);
    const int rc = submit_and_wait_inner(c, r);
    if (rc != PG_OK) c->outstanding.fetch_sub(1, std::memory_order_relaxed);     // (finished calls leave through finish_call)
    return rc;
}
int submit_and_wait_inner(pg_coalescer* c, Req* r) {
    r->arrived = Clock::now();
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (c->stop || c->broken) {
            if (c->broken) set_error("pg_coalescer: the device failed earlier (%s); re-create the context", c->broken_msg.c_str());
            else set_error("pg_coalescer: already shut down");
            const int rc = c->broken ? PG_ERR_DEVICE : PG_ERR_INVALID;
            delete r;
            return rc;
        }
        c->queue[r->queue].push_back(r);
        if (r->queue >= kQRank0) c->queue_items[r->queue] += r->n;
        const int f = r->queue;
        if (c->seen_arrival[f]) {
            double gap = std::chrono::duration<double, std::micro>(r->arrived - c->last_arrival[f]).count();
            gap = gap < 0.0 ? 0.0 : (gap > 1e5 ? 1e5 : gap);
            c->gap_ewma_us[f] = 0.75 * c->gap_ewma_us[f] + 0.25 * gap;
        } else {
            c->seen_arrival[f] = true;
            c->gap_ewma_us[f] = 0.0;                   // (the first request waits: nothing is known yet)
        }
        c->last_arrival[f] = r->arrived;
    }
    c->cv_dispatch.notify_one();
    const bool timed = c->timeout_us != 0;
    const auto deadline = r->arrived + std::chrono::microseconds(c->timeout_us);
    for (;;) {
        const uint32_t st = r->state.load(std::memory_order_acquire);
        if (st == kDone) return PG_OK;
        if (!timed) {
            futex_wait(&r->state, st, nullptr);
            continue;
        }
        const auto now = Clock::now();
        if (now < deadline) {
            const auto ns = std::chrono::duration_cast<std::chrono::nanoseconds>(deadline - now).count();
            timespec ts;
            ts.tv_sec = (time_t)(ns / 1000000000);
            ts.tv_nsec = (long)(ns % 1000000000);
            futex_wait(&r->state, st, &ts);
            continue;
        }
        // deadline: leave.  Still queued → take the request back; being staged → wait the few microseconds that
        // takes; staged / in flight → abandon the record to the workers.
        {
            std::lock_guard<std::mutex> g(c->mu);
            if (r->state.load(std::memory_order_acquire) == kQueued) {
                auto& q = c->queue[r->queue];
                auto it = std::find(q.begin(), q.end(), r);
                if (it != q.end()) {                   // (always, while the state is kQueued: both change under c->mu)
                    q.erase(it);
                    if (r->queue >= kQRank0) c->queue_items[r->queue] -= r->n;
                    c->stats.timeouts++;
                    delete r;
                    set_error("pg_coalescer: deadline of %u us passed while the request was queued", c->timeout_us);
                    return PG_ERR_TIMEOUT;
                }
            }
        }
        while (r->state.load(std::memory_order_acquire) == kStaging) std::this_thread::yield();
        uint32_t expect = kStaged;
        if (r->state.compare_exchange_strong(expect, kAbandoned, std::memory_order_acq_rel)) {
            {
                std::lock_guard<std::mutex> g(c->mu);
                c->stats.timeouts++;
            }
            req_unref(r);
            set_error("pg_coalescer: deadline of %u us passed while the request's batch was on the device", c->timeout_us);
            return PG_ERR_TIMEOUT;
        }
        // (finished in the meantime)
    }
}

// what every entry point does after submit_and_wait returned PG_OK and the slice was copied
int finish_call(pg_coalescer* c, Req* r) {
    c->outstanding.fetch_sub(1, std::memory_order_relaxed);
    Slot* s = r->slot;
    const int rc = r->rc;
    if (rc != PG_OK) set_error("%s", r->err);
    if (s) release_slot(c, s);
    req_unref(r);
    return rc;
}

void destroy_partial(pg_coalescer* c) {
    for (Slot* o : c->slots) free_slot(c, o);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->sibling) pg_shutdown(c->sibling);
    delete c;
}

}  // namespace
}  // namespace pg

extern "C" {

int pg_coalescer_create_scene(pg_ctx* ctx, const pg_table* t, const pg_scene_config* sc, pg_coalescer** out) {
    PG_REQUIRE(ctx && t && sc && out, "pg_coalescer_create: NULL argument");
    const pg_coalescer_config* cfg = &sc->base;
    PG_REQUIRE(cfg->k >= 1 && cfg->k <= 16384, "pg_coalescer_create: k=%u unsupported (1..16384)", cfg->k);
    const uint32_t max_q = t->dim <= 128 ? (uint32_t)pg::kMaxQueries : 32u;
    PG_REQUIRE(cfg->max_batch <= max_q, "pg_coalescer_create: max_batch %u exceeds %u queries per pass at dim %u",
               cfg->max_batch, max_q, t->dim);
    PG_REQUIRE(cfg->depth <= 4, "pg_coalescer_create: depth %u (at most 4)", cfg->depth);
    PG_REQUIRE(cfg->max_top_n <= cfg->k, "pg_coalescer_create: max_top_n %u exceeds k %u", cfg->max_top_n, cfg->k);
    PG_REQUIRE(sc->n_algos <= (uint32_t)pg::kMaxAlgos && (sc->n_algos == 0 || sc->algos), "pg_coalescer_create: at most %d rank algorithms", pg::kMaxAlgos);
    PG_REQUIRE(!sc->rank_score || sc->n_algos > 0, "pg_coalescer_create: a RankScore expression needs at least one rank algorithm");
    uint32_t d_user_max = 0, nuf_max = 0;
    for (uint32_t a = 0; a < sc->n_algos; ++a) {
        const pg_rank_algo& al = sc->algos[a];
        PG_REQUIRE(al.model && al.name && al.name[0], "pg_coalescer_create: rank algorithm %u needs a model and a name", a);
        if (al.model->kind == PG_MODEL_DNN3) {
            PG_REQUIRE(al.model->d_item == t->dim, "pg_coalescer_create: DNN3 algorithm \"%s\" must rank the table's rows (d_item %u, dim %u)",
                       al.name, al.model->d_item, t->dim);
        } else {
            PG_REQUIRE((al.features && al.item_field_cols) || al.item_rows,
                       "pg_coalescer_create: FM + two-tower algorithm \"%s\" needs its feature columns or its materialised item records", al.name);
            PG_REQUIRE(!al.item_rows || al.item_rows->m == al.model, "pg_coalescer_create: \"%s\": the item records belong to another model", al.name);
            PG_REQUIRE(al.model->nif <= 16, "pg_coalescer_create: \"%s\": at most 16 item fields", al.name);
            nuf_max = std::max(nuf_max, al.model->nuf);
        }
        PG_REQUIRE(!sc->rank_score || al.model->d_user == t->dim,
                   "pg_coalescer_create: recommend needs d_user = the table's dim (the user vector is the query); \"%s\" has %u", al.name,
                   al.model->d_user);
        d_user_max = std::max(d_user_max, al.model->d_user);
    }
    if (sc->query_model) {
        PG_REQUIRE(sc->query_model->kind == PG_MODEL_FM_TWOTOWER && sc->query_model->to == t->dim,
                   "pg_coalescer_create: the query model must be FM_TWOTOWER with tower output = the table's dim");
    }
    const pg_table* trig = sc->trigger_table ? sc->trigger_table : t;
    PG_REQUIRE(trig->dim == t->dim, "pg_coalescer_create: the trigger table's dim %u differs from the table's %u", trig->dim, t->dim);
    PG_REQUIRE(sc->rerank == 0 || sc->rerank == 1, "pg_coalescer_create: rerank must be 0 (none) or 1 (DPPSort)");
    const uint32_t max_top_n = cfg->max_top_n ? cfg->max_top_n : cfg->k;
    if (sc->rerank) {
        PG_REQUIRE(sc->rank_score, "pg_coalescer_create: the DPP stage sits behind a RankScore sort");
        PG_REQUIRE(sc->rerank_candidates >= 1 && sc->rerank_candidates <= std::min<uint32_t>(cfg->k, 1024),
                   "pg_coalescer_create: rerank_candidates %u outside 1..min(k, 1024)", sc->rerank_candidates);
        PG_REQUIRE(max_top_n <= sc->rerank_candidates, "pg_coalescer_create: max_top_n %u exceeds rerank_candidates %u (DPP candidates = max(ctx.Size, CandidateCount) "
                   "must not depend on the request)", max_top_n, sc->rerank_candidates);
        PG_REQUIRE(sc->dpp.norm_relevance_score >= 0 && sc->dpp.norm_relevance_score <= 2, "pg_coalescer_create: dpp.norm_relevance_score must be 0, 1 or 2");
    }
    PG_HIP(hipSetDevice(ctx->device));
    pg_coalescer* c = new pg_coalescer();
    c->ctx = ctx;
    c->t = t;
    c->trigger_table = trig;
    c->query_model = sc->query_model;
    c->n_algos = (int)sc->n_algos;
    // score planes and their RankScore names: "<algo>" or, per output of a multi-output model, "<algo>_<output>"
    // (rank_service.go:315-319)
    const char* names[pg::kMaxPlanes] = {nullptr};
    for (int a = 0; a < c->n_algos; ++a) {
        const uint32_t heads = sc->algos[a].model->n_out;
        c->plane0[a] = c->n_planes;
        // (for EVERY algorithm, before its names are pushed: a single-output one behind models that already fill the
        // planes would write past names[] below — ADVICE r4)
        if (c->n_planes + (int)heads > pg::kMaxPlanes) {
            pg::set_error("pg_coalescer_create: the scene's models have more than %d outputs together", pg::kMaxPlanes);
            delete c;
            return PG_ERR_INVALID;
        }
        if (heads > 1) {
            for (uint32_t o = 0; o < heads; ++o) {
                // (no names given: "<algo>_0", "<algo>_1", … — enough for a scene that only ranks)
                const char* on = sc->algos[a].output_names ? sc->algos[a].output_names[o] : nullptr;
                if (sc->algos[a].output_names && (!on || !on[0])) {
                    pg::set_error("pg_coalescer_create: \"%s\": output %u has no name", sc->algos[a].name, o);
                    delete c;
                    return PG_ERR_INVALID;
                }
                c->plane_names.push_back(std::string(sc->algos[a].name) + "_" + (on ? std::string(on) : std::to_string(o)));
            }
        } else {
            c->plane_names.push_back(sc->algos[a].name);
        }
        c->n_planes += (int)heads;
        c->max_heads = std::max(c->max_heads, heads);
    }
    for (int p_ = 0; p_ < c->n_planes; ++p_) names[p_] = c->plane_names[(size_t)p_].c_str();
    for (int a = 0; a < c->n_algos; ++a) {
        c->algos[a].m = sc->algos[a].model;
        c->algos[a].fs = sc->algos[a].features;
        c->algos[a].irows = sc->algos[a].item_rows;
        if (sc->algos[a].item_field_cols)
            for (uint32_t f = 0; f < sc->algos[a].model->nif && f < 16; ++f) c->algos[a].item_field_cols[f] = sc->algos[a].item_field_cols[f];
        c->algo_names[a] = sc->algos[a].name;
    }
    c->e = sc->rank_score;
    c->e_hold.take(c->e);
    int rc;
    if (c->e && (rc = pg::recommend_bind_vars(c->e, names, c->n_planes, &c->var_src, "pg_coalescer_create"))) {
        delete c;
        return rc;
    }
    c->rerank.kind = sc->rerank;
    c->rerank.candidates = sc->rerank_candidates;
    c->rerank.dpp = sc->dpp;
    c->k = cfg->k;
    c->max_batch = cfg->max_batch ? cfg->max_batch : max_q;
    c->max_wait_us = cfg->max_wait_us ? cfg->max_wait_us : 100;
    for (double& sc : c->rejoin_score) sc = 1.0;     // (optimistic: callers are closed loops until they show otherwise)
    c->depth = cfg->depth ? cfg->depth : 2;
    c->max_top_n = max_top_n;
    c->max_rank_items = cfg->max_rank_items ? cfg->max_rank_items : cfg->k;
    c->timeout_us = cfg->timeout_us;
    c->dim = t->dim;
    c->vec_w = std::max(t->dim, d_user_max);
    c->ufid_stride = nuf_max;
    // a rank batch: as many candidates as a full recommend batch ranks, from at most 16384 calls
    c->rank_item_cap = std::max<uint32_t>(c->max_batch * c->k, c->max_rank_items);
    c->max_rank_reqs = 16384;
    c->vec_rows = c->n_algos ? std::max<uint32_t>(c->max_batch, c->max_rank_reqs) : c->max_batch;
    c->rank_stride = std::max<size_t>((size_t)c->max_batch * c->k, c->rank_item_cap);
    c->dpp_max_n = sc->max_rerank_items ? sc->max_rerank_items : 1024;
    c->dpp_max_hook = sc->max_hook_dim;
    c->dpp_item_cap = std::max<uint32_t>(c->dpp_max_n, 256u * 512u);
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        pg::set_error("pg_coalescer_create: %s", hipGetErrorString(hipGetLastError()));
        c->copy_stream = nullptr;
        pg::destroy_partial(c);
        return PG_ERR_DEVICE;
    }
    { const char* tv = getenv("PG_COALESCER_TIMERS"); c->timers = tv && atoi(tv) != 0; }
    if (c->depth >= 2 && pg_init(ctx->device, nullptr, &c->sibling) != PG_OK) c->sibling = nullptr;   // (optional: one stream works too)
    if (c->sibling) {
        // the sibling serves every other batch: it must plan exactly as the caller's context does
        std::lock_guard<std::mutex> g(ctx->mu);
        c->sibling->knobs = ctx->knobs;
    }
    for (uint32_t i = 0; i < c->depth; ++i) {
        pg::Slot* s = new pg::Slot();
        s->id = (int)i;
        s->ctx = (c->sibling && (i & 1)) ? c->sibling : ctx;
        if ((rc = pg::alloc_slot(c, s))) {
            pg::free_slot(c, s);
            pg::destroy_partial(c);
            return rc;
        }
        c->slots.push_back(s);
        c->free_slots.push_back(s);
    }
    // build the table's shadow now rather than inside the first batch
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        if ((rc = pg::ensure_table_stats(ctx, t))) {
            pg::destroy_partial(c);
            return rc;
        }
    }
    c->dispatcher = std::thread(pg::dispatcher_main, c);
    c->completer = std::thread(pg::completer_main, c);
    *out = c;
    return PG_OK;
}

int pg_coalescer_create(pg_ctx* ctx, const pg_table* t, const pg_model* m, const pg_expr* e, const char* rank_var,
                        const pg_coalescer_config* cfg, pg_coalescer** out) {
    PG_REQUIRE(ctx && t && cfg && out, "pg_coalescer_create: NULL argument");
    PG_REQUIRE(!e || (m && rank_var), "pg_coalescer_create: a RankScore expression needs a model and its name");
    PG_REQUIRE(!m || m->kind == PG_MODEL_DNN3, "pg_coalescer_create: the model must be DNN3 over the table's rows (other scenes: pg_coalescer_create_scene)");
    pg_scene_config sc;
    memset(&sc, 0, sizeof sc);
    sc.base = *cfg;
    pg_rank_algo al;
    memset(&al, 0, sizeof al);
    al.model = m;
    al.name = rank_var ? rank_var : "rank";
    if (m) {
        sc.algos = &al;
        sc.n_algos = 1;
    }
    sc.rank_score = e;
    return pg_coalescer_create_scene(ctx, t, &sc, out);
}

int pg_coalescer_destroy(pg_coalescer* c) {
    if (!c) return PG_OK;
    {
        std::lock_guard<std::mutex> g(c->mu);
        c->stop = true;
    }
    c->cv_dispatch.notify_all();
    c->cv_complete.notify_all();
    if (c->dispatcher.joinable()) c->dispatcher.join();
    if (c->completer.joinable()) c->completer.join();
    // callers of the last batches may still be copying their slices
    for (pg::Slot* s : c->slots)
        while (s->pending.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    hipSetDevice(c->ctx->device);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    if (!c->group) hipStreamSynchronize(c->ctx->stream);
    if (c->sibling) hipStreamSynchronize(c->sibling->stream);
    for (pg::Slot* s : c->slots) pg::free_slot(c, s);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->sibling) pg_shutdown(c->sibling);
    delete c;
    return PG_OK;
}

static int coalescer_recall_common(pg_coalescer* c, pg::Req* r, uint64_t* out_rows, float* out_scores, uint32_t* out_count) {
    if (c->group) {
        delete r;
        pg::set_error("pg_coalescer: a coalescer over a shard group serves pg_coalescer_recommend only");
        return PG_ERR_UNSUPPORTED;
    }
    r->queue = pg::kQRecall;
    int rc;
    if ((rc = pg::submit_and_wait(c, r))) return rc;
    pg::Slot* s = r->slot;
    if (r->rc == PG_OK) {
        const size_t k = c->k;
        memcpy(out_rows, s->h_out + (size_t)r->index * k * 8, k * 8);
        memcpy(out_scores, s->h_out + (size_t)c->max_batch * k * 8 + (size_t)r->index * k * 4, k * 4);
        if (out_count) *out_count = s->run->h_status[1 + r->index];
    }
    return pg::finish_call(c, r);
}

int pg_coalescer_recall(pg_coalescer* c, const float* query, uint64_t* out_rows, float* out_scores,
                        uint32_t* out_count) {
    PG_REQUIRE(c && query && out_rows && out_scores, "pg_coalescer_recall: NULL argument");
    pg::Req* r = new pg::Req();
    r->vec = query;
    r->qkind = pg::kVector;
    return coalescer_recall_common(c, r, out_rows, out_scores, out_count);
}

int pg_coalescer_recall_l2(pg_coalescer* c, const float* query, uint64_t* out_rows, float* out_dist, uint32_t* out_count) {
    PG_REQUIRE(c && query && out_rows && out_dist, "pg_coalescer_recall_l2: NULL argument");
    PG_REQUIRE(c->dim == 64 || c->dim == 128, "pg_coalescer_recall_l2: dim %u unsupported (64 or 128)", c->dim);
    if (c->group) {
        pg::set_error("pg_coalescer: a coalescer over a shard group serves pg_coalescer_recommend only");
        return PG_ERR_UNSUPPORTED;
    }
    pg::Req* r = new pg::Req();
    r->vec = query;
    r->qkind = pg::kVector;
    r->queue = pg::kQRecallL2;
    int rc;
    if ((rc = pg::submit_and_wait(c, r))) return rc;
    pg::Slot* s = r->slot;
    if (r->rc == PG_OK) {
        const size_t k = c->k;
        memcpy(out_rows, s->h_out + (size_t)r->index * k * 8, k * 8);
        memcpy(out_dist, s->h_out + (size_t)c->max_batch * k * 8 + (size_t)r->index * k * 4, k * 4);
        if (out_count) *out_count = s->run->h_status[1 + r->index];
    }
    return pg::finish_call(c, r);
}

int pg_coalescer_i2i_recall(pg_coalescer* c, uint32_t trigger_row, uint64_t* out_rows, float* out_scores,
                            uint32_t* out_count) {
    PG_REQUIRE(c && out_rows && out_scores, "pg_coalescer_i2i_recall: NULL argument");
    PG_REQUIRE(!c->trigger_table->d_row_map, "pg_coalescer_i2i_recall: the trigger table is a filtered view (trigger rows are rows of the source)");
    PG_REQUIRE(trigger_row < c->trigger_table->rows, "pg_coalescer_i2i_recall: trigger row %u outside table of %llu rows", trigger_row,
               (unsigned long long)c->trigger_table->rows);
    pg::Req* r = new pg::Req();
    r->qkind = pg::kTrigger;
    r->trigger_row = trigger_row;
    return coalescer_recall_common(c, r, out_rows, out_scores, out_count);
}

int pg_coalescer_online_recall(pg_coalescer* c, const float* user_vec, uint64_t* out_rows, float* out_scores,
                               uint32_t* out_count) {
    PG_REQUIRE(c && user_vec && out_rows && out_scores, "pg_coalescer_online_recall: NULL argument");
    PG_REQUIRE(c->query_model, "pg_coalescer_online_recall: the scene has no query model");
    pg::Req* r = new pg::Req();
    r->vec = user_vec;
    r->qkind = pg::kOnline;
    return coalescer_recall_common(c, r, out_rows, out_scores, out_count);
}

int pg_coalescer_rank(pg_coalescer* c, uint32_t algo, const float* user_vec, const int32_t* user_field_ids,
                      const uint32_t* cand_rows, uint32_t n, float* out_scores) {
    PG_REQUIRE(c && user_vec, "pg_coalescer_rank: NULL argument");
    PG_REQUIRE(algo < (uint32_t)c->n_algos, "pg_coalescer_rank: algorithm %u of %d", algo, c->n_algos);
    PG_REQUIRE(n <= c->max_rank_items, "pg_coalescer_rank: %u candidates exceed max_rank_items %u", n, c->max_rank_items);
    if (n == 0) return PG_OK;
    PG_REQUIRE(cand_rows && out_scores, "pg_coalescer_rank: NULL argument");
    const pg_model* m = c->algos[algo].m;
    if (m->kind == PG_MODEL_DNN3) {
        for (uint32_t i = 0; i < n; ++i)
            PG_REQUIRE(cand_rows[i] < c->t->rows, "pg_coalescer_rank: candidate %u row %u outside table of %llu rows", i,
                       cand_rows[i], (unsigned long long)c->t->rows);
    } else {
        PG_REQUIRE(user_field_ids, "pg_coalescer_rank: an FM + two-tower algorithm needs the user's field ids");
        for (uint32_t f = 0; f < m->nuf; ++f)
            PG_REQUIRE(user_field_ids[f] >= 0 && (uint32_t)user_field_ids[f] < m->vocab, "pg_coalescer_rank: user field id %d outside vocab %u",
                       user_field_ids[f], m->vocab);
    }
    pg::Req* r = new pg::Req();
    r->queue = pg::kQRank0 + (int)algo;
    r->vec = user_vec;
    r->ufids = user_field_ids;
    r->cand = cand_rows;
    r->n = n;
    int rc;
    if ((rc = pg::submit_and_wait(c, r))) return rc;
    if (r->rc == PG_OK)                                  // head o: out_scores[o * n ..), from plane o of the batch image
        for (uint32_t o = 0; o < m->n_out; ++o)
            memcpy(out_scores + (size_t)o * n, r->slot->h_out + ((size_t)o * r->slot->n_items + r->item0) * 4, (size_t)n * 4);
    return pg::finish_call(c, r);
}

int pg_coalescer_rank_dnn3(pg_coalescer* c, const float* user_vec, const uint32_t* cand_rows, uint32_t n,
                           float* out_scores) {
    PG_REQUIRE(c, "pg_coalescer_rank_dnn3: NULL argument");
    for (int a = 0; a < c->n_algos; ++a)
        if (c->algos[a].m->kind == PG_MODEL_DNN3) return pg_coalescer_rank(c, (uint32_t)a, user_vec, nullptr, cand_rows, n, out_scores);
    pg::set_error("pg_coalescer_rank_dnn3: the coalescer was created without a DNN3 model");
    return PG_ERR_INVALID;
}

int pg_coalescer_rank_fm2t(pg_coalescer* c, const float* user_vec, const int32_t* user_field_ids,
                           const uint32_t* cand_rows, uint32_t n, float* out_scores) {
    PG_REQUIRE(c, "pg_coalescer_rank_fm2t: NULL argument");
    for (int a = 0; a < c->n_algos; ++a)
        if (c->algos[a].m->kind == PG_MODEL_FM_TWOTOWER)
            return pg_coalescer_rank(c, (uint32_t)a, user_vec, user_field_ids, cand_rows, n, out_scores);
    pg::set_error("pg_coalescer_rank_fm2t: the scene has no FM + two-tower algorithm");
    return PG_ERR_INVALID;
}

static int coalescer_recommend_common(pg_coalescer* c, const float* user_vec, const int32_t* user_field_ids, uint32_t top_n,
                                      uint64_t* out_rows, float* out_recall_scores, float* out_rank_scores, int rank_planes,
                                      double* out_fused, uint32_t* out_count) {
    PG_REQUIRE(c && user_vec && out_rows && out_recall_scores && out_rank_scores && out_fused,
               "pg_coalescer_recommend: NULL argument");
    PG_REQUIRE(c->e, "pg_coalescer_recommend: the coalescer was created without a RankScore expression");
    PG_REQUIRE(top_n >= 1 && top_n <= c->max_top_n, "pg_coalescer_recommend: top_n %u outside 1..%u", top_n, c->max_top_n);
    if (c->ufid_stride) {
        PG_REQUIRE(user_field_ids, "pg_coalescer_recommend: the scene ranks with an FM + two-tower algorithm: pass the user's field ids (pg_coalescer_recommend_ex)");
        for (int a = 0; a < c->n_algos; ++a) {
            const pg_model* m = c->algos[a].m;
            if (m->kind != PG_MODEL_FM_TWOTOWER) continue;
            for (uint32_t f = 0; f < m->nuf; ++f)
                PG_REQUIRE(user_field_ids[f] >= 0 && (uint32_t)user_field_ids[f] < m->vocab, "pg_coalescer_recommend: user field id %d outside vocab %u",
                           user_field_ids[f], m->vocab);
        }
    }
    pg::Req* r = new pg::Req();
    r->queue = pg::kQRecommend;
    r->vec = user_vec;
    r->ufids = user_field_ids;
    r->n = top_n;
    int rc;
    if ((rc = pg::submit_and_wait(c, r))) return rc;
    pg::Slot* s = r->slot;
    if (r->rc == PG_OK && c->group) {
        const size_t o = (size_t)r->index * s->n_items;                 // the step's pages are [n_req][batch top_n]
        memcpy(out_rows, s->g_rows + o, (size_t)top_n * 8);
        memcpy(out_fused, s->g_fus + o, (size_t)top_n * 8);
        memcpy(out_recall_scores, s->g_rec + o, (size_t)top_n * 4);
        memcpy(out_rank_scores, s->g_rnk + o, (size_t)top_n * 4);
        if (out_count) *out_count = std::min(top_n, s->g_cnt[r->index]);
    } else if (r->rc == PG_OK) {
        const size_t nq_top = (size_t)s->n_items;                       // page width of the batch's image: planes are [n_req][nq_top]
        const uint32_t* counts = (const uint32_t*)(s->h_out + pg::page_bytes(c));
        const uint32_t* pick_counts = counts + c->max_batch;
        const uint32_t batch = s->n_req;
        const size_t np = (size_t)batch * nq_top;
        const uint64_t* p_rows = (const uint64_t*)s->h_out;
        const double* p_fused = (const double*)(p_rows + np);
        const float* p_recall = (const float*)(p_fused + np);
        const float* p_rank = p_recall + np;
        const size_t o = (size_t)r->index * nq_top;
        memcpy(out_rows, p_rows + o, (size_t)top_n * 8);
        memcpy(out_fused, p_fused + o, (size_t)top_n * 8);
        memcpy(out_recall_scores, p_recall + o, (size_t)top_n * 4);
        for (int a = 0; a < rank_planes; ++a) memcpy(out_rank_scores + (size_t)a * top_n, p_rank + (size_t)a * np + o, (size_t)top_n * 4);
        if (out_count) {
            uint32_t cnt = std::min(top_n, counts[r->index]);
            if (c->rerank.kind) cnt = std::min(cnt, pick_counts[r->index]);
            *out_count = cnt;
        }
    }
    return pg::finish_call(c, r);
}

int pg_coalescer_recommend(pg_coalescer* c, const float* user_vec, uint32_t top_n, uint64_t* out_rows,
                           float* out_recall_scores, float* out_rank_scores, double* out_fused,
                           uint32_t* out_count) {
    return coalescer_recommend_common(c, user_vec, nullptr, top_n, out_rows, out_recall_scores, out_rank_scores, 1, out_fused, out_count);
}

int pg_coalescer_recommend_ex(pg_coalescer* c, const float* user_vec, const int32_t* user_field_ids, uint32_t top_n,
                              uint64_t* out_rows, float* out_recall_scores, float* out_rank_scores, double* out_fused,
                              uint32_t* out_count) {
    PG_REQUIRE(c, "pg_coalescer_recommend_ex: NULL argument");
    return coalescer_recommend_common(c, user_vec, user_field_ids, top_n, out_rows, out_recall_scores, out_rank_scores, c->n_planes, out_fused,
                                      out_count);
}

int pg_coalescer_dpp(pg_coalescer* c, const uint32_t* cand_rows, const double* rel, uint32_t n,
                     const pg_dpp_options* o, const double* hook_emb, uint32_t* out_idx, uint32_t* out_count,
                     double* out_relevance) {
    PG_REQUIRE(c && o && out_count, "pg_coalescer_dpp: NULL argument");
    PG_REQUIRE(!c->group, "pg_coalescer_dpp: a coalescer over a shard group serves pg_coalescer_recommend only");
    *out_count = 0;
    if (n == 0 || o->topn == 0) return PG_OK;
    PG_REQUIRE(rel && out_idx, "pg_coalescer_dpp: NULL argument");
    PG_REQUIRE(o->norm_relevance_score >= 0 && o->norm_relevance_score <= 2, "pg_coalescer_dpp: norm_relevance_score must be 0, 1 or 2");
    PG_REQUIRE(!o->has_table || cand_rows, "pg_coalescer_dpp: has_table needs candidate rows");
    PG_REQUIRE(o->has_table || o->hook_dim > 0, "pg_coalescer_dpp: no embedding table and no hook embeddings (the reference returns the items unchanged)");
    PG_REQUIRE(o->hook_dim == 0 || hook_emb, "pg_coalescer_dpp: hook_dim > 0 but hook_emb is NULL");
    if (n > c->dpp_max_n || o->hook_dim > c->dpp_max_hook) {
        pg::set_error("pg_coalescer_dpp: %u candidates / hook width %u exceed the scene's max_rerank_items %u / max_hook_dim %u", n, o->hook_dim,
                      c->dpp_max_n, c->dpp_max_hook);
        return PG_ERR_UNSUPPORTED;
    }
    if (o->has_table)
        for (uint32_t i = 0; i < n; ++i)
            PG_REQUIRE(cand_rows[i] < c->t->rows, "pg_coalescer_dpp: candidate row %u outside table", cand_rows[i]);
    pg::Req* r = new pg::Req();
    r->rel.resize(n);
    // dpp_norm_relevance_score: O(n) scalar work on the caller's thread, as pg_dpp_ex does it
    if (!pg::dpp_norm_relevance_host(rel, n, o->norm_relevance_score, r->rel.data())) {
        delete r;
        pg::set_error("pg_dpp: all item score is zero (dpp_sort.go:385-397); the caller keeps the items unchanged");
        return PG_ERR_ARITH;
    }
    if (out_relevance) memcpy(out_relevance, r->rel.data(), (size_t)n * 8);      // "dpp_relevance_score" (:410)
    r->queue = pg::kQDpp;
    r->cand = cand_rows;
    r->hook = hook_emb;
    r->n = n;
    r->key.n = n;
    r->key.topn = std::min(o->topn, n);
    r->key.window = o->window ? o->window : 10;
    r->key.hook_dim = o->hook_dim;
    r->key.normalize = o->normalize_emb ? 1 : 0;
    r->key.ensure_pos = o->ensure_pos_similarity ? 1 : 0;
    r->key.has_table = o->has_table ? 1 : 0;
    r->key.alpha = o->alpha;
    int rc;
    if ((rc = pg::submit_and_wait(c, r))) return rc;
    pg::Slot* s = r->slot;
    if (r->rc == PG_OK) {
        const uint32_t topn = s->key.topn;
        const uint32_t cnt = std::min(topn, s->h_dout[c->dpp_item_cap + r->index]);
        memcpy(out_idx, s->h_dout + (size_t)r->index * topn, (size_t)cnt * 4);
        *out_count = cnt;
    }
    return pg::finish_call(c, r);
}

int pg_coalescer_ssd(pg_coalescer* c, const uint32_t* cand_rows, const double* rel, uint32_t n, double gamma, uint32_t topn,
                     uint32_t window, int normalize_emb, int ensure_pos_similarity, int norm_quality_score, int use_ssd_star,
                     uint32_t* out_idx, uint32_t* out_count, double* out_quality) {
    PG_REQUIRE(c && out_count, "pg_coalescer_ssd: NULL argument");
    PG_REQUIRE(!c->group, "pg_coalescer_ssd: a coalescer over a shard group serves pg_coalescer_recommend only");
    *out_count = 0;
    if (n == 0 || topn == 0) return PG_OK;
    PG_REQUIRE(cand_rows && rel && out_idx, "pg_coalescer_ssd: NULL argument");
    PG_REQUIRE(norm_quality_score >= 0 && norm_quality_score <= 2, "pg_coalescer_ssd: norm_quality_score must be 0, 1 or 2");
    if (window <= 1) window = 5;                          // ssd_sort.go:357-360
    const uint32_t d1 = c->dim + (ensure_pos_similarity ? 1u : 0u);
    if (n > c->dpp_max_n || !pg::ssd_batchable(d1, window)) {
        pg::set_error("pg_coalescer_ssd: %u candidates (max_rerank_items %u) / dim %u / window %u: not a batchable shape (pg_ssd serves it)", n,
                      c->dpp_max_n, c->dim, window);
        return PG_ERR_UNSUPPORTED;
    }
    for (uint32_t i = 0; i < n; ++i)
        PG_REQUIRE(cand_rows[i] < c->t->rows, "pg_coalescer_ssd: candidate row %u outside table", cand_rows[i]);
    pg::Req* r = new pg::Req();
    r->rel.resize(n);
    // ssd_norm_quality_score: O(n) scalar work on the caller's thread, as pg_ssd does it
    if (!pg::ssd_norm_quality_host(rel, n, norm_quality_score, r->rel.data())) {
        delete r;                                         // "all item score are zeros": the items stay as they are
        for (uint32_t i = 0; i < n; ++i) out_idx[i] = i;
        *out_count = n;
        if (out_quality) memcpy(out_quality, rel, (size_t)n * 8);
        return PG_OK;
    }
    if (out_quality) memcpy(out_quality, r->rel.data(), (size_t)n * 8);
    r->queue = pg::kQDpp;
    r->cand = cand_rows;
    r->n = n;
    r->key.n = n;
    r->key.topn = std::min(topn, n);
    r->key.window = window;
    r->key.normalize = normalize_emb ? 1 : 0;
    r->key.ensure_pos = ensure_pos_similarity ? 1 : 0;
    r->key.has_table = 1;
    r->key.ssd = 1;
    r->key.star = use_ssd_star ? 1 : 0;
    r->key.alpha = gamma;
    int rc;
    if ((rc = pg::submit_and_wait(c, r))) return rc;
    pg::Slot* s = r->slot;
    if (r->rc == PG_OK) {
        const uint32_t T = s->key.topn;
        memcpy(out_idx, s->h_dout + (size_t)r->index * T, (size_t)T * 4);
        *out_count = T;
    }
    return pg::finish_call(c, r);
}

int pg_coalescer_stats(pg_coalescer* c, pg_coalescer_stats_t* out) {
    PG_REQUIRE(c && out, "pg_coalescer_stats: NULL argument");
    std::lock_guard<std::mutex> g(c->mu);
    *out = c->stats;
    return PG_OK;
}

int pg_debug_stall(pg_ctx* ctx, uint32_t ms) {
    PG_REQUIRE(ctx, "pg_debug_stall: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    int khz = 0;
    PG_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    if (khz <= 0) khz = 100000;
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::stall_kernel<<<1, 64, 0, ctx->stream>>>((uint64_t)ms * (uint64_t)khz);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

int pg_coalescer_create_group(pg_group* g, const pg_expr* e, const char* rank_var, const pg_group_plan* plan,
                              const pg_coalescer_config* cfg, pg_coalescer** out) {
    PG_REQUIRE(g && e && rank_var && plan && cfg && out, "pg_coalescer_create_group: NULL argument");
    uint64_t total = 0;
    uint32_t dim = 0;
    pg_group_info(g, &total, &dim);
    PG_REQUIRE(total > 0 && pg_group_ctx(g, 0), "pg_coalescer_create_group: the group has no table");
    PG_REQUIRE(plan->k >= 1 && plan->k <= 16384 && (cfg->k == 0 || cfg->k == plan->k), "pg_coalescer_create_group: k comes from the plan");
    PG_REQUIRE(cfg->max_batch <= (uint32_t)pg::kMaxQueries, "pg_coalescer_create_group: max_batch %u exceeds %d", cfg->max_batch, pg::kMaxQueries);
    PG_REQUIRE(cfg->depth <= 2, "pg_coalescer_create_group: depth %u (a group runs at most two steps at a time)", cfg->depth);
    PG_REQUIRE(cfg->max_top_n <= plan->k, "pg_coalescer_create_group: max_top_n %u exceeds k %u", cfg->max_top_n, plan->k);
    const char* names[1] = {rank_var};
    std::vector<int> src;
    int rc;
    if ((rc = pg::recommend_bind_vars(e, names, 1, &src, "pg_coalescer_create_group"))) return rc;
    pg_coalescer* c = new pg_coalescer();
    c->group = g;
    c->gplan = *plan;
    c->grank_var = rank_var;
    c->ctx = pg_group_ctx(g, 0);
    c->e = e;
    c->e_hold.take(e);
    c->k = plan->k;
    c->max_batch = cfg->max_batch ? cfg->max_batch : (uint32_t)pg::kMaxQueries;
    c->max_wait_us = cfg->max_wait_us ? cfg->max_wait_us : 100;
    for (double& sc : c->rejoin_score) sc = 1.0;     // (optimistic: callers are closed loops until they show otherwise)
    c->depth = cfg->depth ? cfg->depth : 2;
    c->max_top_n = cfg->max_top_n ? cfg->max_top_n : std::min<uint32_t>(plan->k, 1000);
    if (plan->dpp_candidates && c->max_top_n > plan->dpp_candidates) c->max_top_n = plan->dpp_candidates;   // DPP candidates must not depend on the request
    c->timeout_us = cfg->timeout_us;
    c->dim = dim;
    PG_HIP(hipSetDevice(c->ctx->device));
    for (uint32_t i = 0; i < c->depth; ++i) {
        pg::Slot* s = new pg::Slot();
        s->id = (int)i;
        s->ctx = c->ctx;
        if ((rc = pg::alloc_slot(c, s))) {
            pg::free_slot(c, s);
            pg::destroy_partial(c);
            return rc;
        }
        c->slots.push_back(s);
        c->free_slots.push_back(s);
    }
    c->dispatcher = std::thread(pg::dispatcher_main, c);
    c->completer = std::thread(pg::completer_main, c);
    *out = c;
    return PG_OK;
}

}  // extern "C"

struct pg_router {
    std::vector<pg_coalescer*> rep;
    std::vector<std::atomic<uint64_t>> served;
    std::atomic<uint32_t> rr{0};
    explicit pg_router(size_t n) : served(n) {}
};

namespace pg {
namespace {
// the replica with the fewest requests outstanding; ties go round robin
uint32_t router_pick(pg_router* r) {
    const uint32_t n = (uint32_t)r->rep.size();
    const uint32_t start = r->rr.fetch_add(1, std::memory_order_relaxed) % n;
    uint32_t best = start, load = r->rep[start]->outstanding.load(std::memory_order_relaxed);
    for (uint32_t i = 1; i < n; ++i) {
        const uint32_t j = (start + i) % n;
        const uint32_t l = r->rep[j]->outstanding.load(std::memory_order_relaxed);
        if (l < load) {
            load = l;
            best = j;
        }
    }
    r->served[best].fetch_add(1, std::memory_order_relaxed);
    return best;
}
}  // namespace
}  // namespace pg

extern "C" {

int pg_router_create(pg_coalescer* const* replicas, uint32_t n, pg_router** out) {
    PG_REQUIRE(replicas && out && n >= 1 && n <= 64, "pg_router_create: bad argument");
    for (uint32_t i = 0; i < n; ++i) {
        PG_REQUIRE(replicas[i], "pg_router_create: replica %u is NULL", i);
        PG_REQUIRE(replicas[i]->k == replicas[0]->k && replicas[i]->dim == replicas[0]->dim && replicas[i]->max_top_n == replicas[0]->max_top_n,
                   "pg_router_create: replica %u differs from replica 0 in k / dim / max_top_n", i);
    }
    pg_router* r = new pg_router(n);
    r->rep.assign(replicas, replicas + n);
    for (auto& s : r->served) s.store(0);
    *out = r;
    return PG_OK;
}

int pg_router_destroy(pg_router* r) {
    delete r;
    return PG_OK;
}

int pg_router_recommend(pg_router* r, const float* user_vec, uint32_t top_n, uint64_t* out_rows,
                        float* out_recall_scores, float* out_rank_scores, double* out_fused, uint32_t* out_count) {
    PG_REQUIRE(r, "pg_router_recommend: NULL argument");
    return pg_coalescer_recommend(r->rep[pg::router_pick(r)], user_vec, top_n, out_rows, out_recall_scores, out_rank_scores, out_fused, out_count);
}

int pg_router_recall(pg_router* r, const float* query, uint64_t* out_rows, float* out_scores, uint32_t* out_count) {
    PG_REQUIRE(r, "pg_router_recall: NULL argument");
    return pg_coalescer_recall(r->rep[pg::router_pick(r)], query, out_rows, out_scores, out_count);
}

int pg_router_stats(pg_router* r, uint64_t* out_served) {
    PG_REQUIRE(r && out_served, "pg_router_stats: NULL argument");
    for (size_t i = 0; i < r->rep.size(); ++i) out_served[i] = r->served[i].load(std::memory_order_relaxed);
    return PG_OK;
}

}  // extern "C"
