// coalescer.hip — cross-request batching inside the library (SURVEY.md 8b "Threading").
//
// pairec calls its plug-ins per request and concurrently: RecallService.GetItems starts one goroutine per recall
// (service/recall.go:129-145), RankService.Rank one per 100-item batch and algorithm (service/rank/rank_service.go:
// 264-289), and HTTP requests overlap.  Every such call used to be one network round trip; here it would be one
// table pass (2.3 ms whether it carries 1 query or 128).  The coalescer turns N concurrent single-request calls
// into one pass:
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
// * Callers sleep in a futex wait on their slot's generation word (a cgo caller parks its OS thread, nothing spins);
//   one FUTEX_WAKE per batch releases them, and every caller copies its own slice out of the slot's pinned output.
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
#include <thread>

namespace pg {
namespace {

using Clock = std::chrono::steady_clock;

enum Flavour { kRecall = 0, kRank = 1, kRecommend = 2 };

struct Slot;

struct Req {
    Flavour kind;
    const float* vec = nullptr;            // query / user vector
    const uint32_t* cand = nullptr;        // rank: candidate rows
    uint32_t n = 0;                        // rank: candidates; recommend: top_n
    Clock::time_point arrived;
    // filled by the workers
    Slot* slot = nullptr;
    uint32_t index = 0;                    // position in the batch
    uint32_t item0 = 0;                    // rank: offset of the request's candidates in the batch
    int rc = PG_OK;
    char err[256] = {0};
    std::atomic<uint32_t> done{0};         // futex word: 0 waiting, 1 finished
};

struct Slot {
    int id = 0;
    Flavour kind = kRecall;
    std::vector<Req*> reqs;
    uint32_t n_req = 0;                    // requests in the batch (reqs is handed back to the callers at wake-up)
    uint32_t n_items = 0;                  // rank: candidates in the batch; recommend: page width of the output image
    PipeRun* run = nullptr;
    pg_ctx* ctx = nullptr;                 // the context (stream + scratch) this slot's batches run on
    hipEvent_t done = nullptr;             // behind the batch's device → host copies
    hipEvent_t computed = nullptr;         // behind its last kernel (the copy stream waits for it)
    Clock::time_point enqueued;
    std::atomic<uint32_t> pending{0};      // callers that have not copied their slice yet
    // pinned host staging
    float* h_vec = nullptr;                // [max_batch][dim] (rank: [max_rank_reqs][d_user])
    uint32_t* h_cand = nullptr;            // rank: concatenated candidate rows
    uint32_t* h_off = nullptr;             // rank: request offsets
    char* h_out = nullptr;                 // flavour-specific output image
    // device
    float* d_vec = nullptr;
    uint32_t* d_cand = nullptr;
    uint32_t* d_off = nullptr;
    uint64_t* d_rows = nullptr;            // [max_batch][k]
    float* d_recall = nullptr;
    float* d_rank = nullptr;               // [max(max_batch * k, rank item capacity)]
    double* d_fused = nullptr;
    uint32_t* d_order = nullptr;
    uint32_t* d_count = nullptr;           // [max_batch]
    char* d_page = nullptr;                // recommend: the sorted pages, layout as h_out
    RecommendCall call;                    // recommend: what was enqueued (the verification may re-run single requests)
};

inline void futex_wait(std::atomic<uint32_t>* w, uint32_t expect) {
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expect, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t>* w) {
    syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

// page[q][j] = entry order[q][j] of request q, j < top_n: the first top_n entries of the sorted list, as four
// planes [nq][top_n] (rows u64 | fused f64 | recall f32 | rank f32)
__global__ void page_gather_kernel(const uint32_t* __restrict__ order, const uint64_t* __restrict__ rows,
                                   const float* __restrict__ recall, const float* __restrict__ rank,
                                   const double* __restrict__ fused, uint32_t nq, uint32_t k, uint32_t top_n,
                                   uint64_t* __restrict__ p_rows, double* __restrict__ p_fused,
                                   float* __restrict__ p_recall, float* __restrict__ p_rank) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * top_n) return;
    const uint32_t q = i / top_n, j = i - q * top_n;
    const size_t src = (size_t)q * k + order[(size_t)q * k + j];
    p_rows[i] = rows[src];
    p_fused[i] = fused[src];
    p_recall[i] = recall[src];
    p_rank[i] = rank[src];
}

}  // namespace
}  // namespace pg

struct pg_coalescer {
    pg_ctx* ctx = nullptr;
    pg_ctx* sibling = nullptr;       // a second context on the same device (own stream and scratch): the slots alternate
                                     // between the two, so the latency-bound head and tail of one batch (pilot, selects,
                                     // fusion, sort) run under the other batch's scan / rank kernels
    const pg_table* t = nullptr;
    const pg_model* m = nullptr;
    const pg_expr* e = nullptr;
    std::vector<int> var_src;
    uint32_t k = 0, max_batch = 0, max_wait_us = 0, depth = 0, max_top_n = 0, max_rank_items = 0;
    uint32_t max_rank_reqs = 0, rank_item_cap = 0;
    uint32_t dim = 0, d_user = 0;
    hipStream_t copy_stream = nullptr;

    std::mutex mu;                                   // queues, slots, stop
    std::condition_variable cv_dispatch;             // new request, slot freed, batch completed
    std::condition_variable cv_complete;             // batch enqueued
    std::deque<pg::Req*> queue[3];
    std::vector<pg::Slot*> slots;
    std::vector<pg::Slot*> free_slots;
    std::deque<pg::Slot*> inflight;
    bool stop = false;
    std::thread dispatcher, completer;
    pg_coalescer_stats_t stats{};
};

namespace pg {
namespace {

size_t page_bytes(const pg_coalescer* c) { return (size_t)c->max_batch * c->max_top_n * 24; }

void fail_req(Req* r, int rc, const char* msg) {
    r->rc = rc;
    snprintf(r->err, sizeof r->err, "%s", msg);
}

int alloc_slot(pg_coalescer* c, Slot* s) {
    const size_t nb = c->max_batch, k = c->k;
    const bool rank = c->m != nullptr;
    const size_t vec_rows = rank ? std::max<size_t>(nb, c->max_rank_reqs) : nb;
    const size_t vec_w = std::max<size_t>(c->dim, c->d_user);
    PG_HIP(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
    PG_HIP(hipEventCreateWithFlags(&s->computed, hipEventDisableTiming));
    PG_HIP(hipHostMalloc((void**)&s->h_vec, vec_rows * vec_w * 4));
    PG_HIP(hipMalloc((void**)&s->d_vec, vec_rows * vec_w * 4));
    size_t out_bytes = nb * k * 12 + nb * 4;                       // recall image: rows | scores | counts
    if (rank) {
        PG_HIP(hipHostMalloc((void**)&s->h_cand, (size_t)c->rank_item_cap * 4));
        PG_HIP(hipHostMalloc((void**)&s->h_off, ((size_t)c->max_rank_reqs + 1) * 4));
        PG_HIP(hipMalloc((void**)&s->d_cand, (size_t)c->rank_item_cap * 4));
        PG_HIP(hipMalloc((void**)&s->d_off, ((size_t)c->max_rank_reqs + 1) * 4));
        out_bytes = std::max(out_bytes, (size_t)c->rank_item_cap * 4);
    }
    if (c->e) out_bytes = std::max(out_bytes, page_bytes(c) + nb * 4);
    PG_HIP(hipHostMalloc((void**)&s->h_out, out_bytes));
    PG_HIP(hipMalloc((void**)&s->d_rows, nb * k * 8));
    PG_HIP(hipMalloc((void**)&s->d_recall, nb * k * 4));
    PG_HIP(hipMalloc((void**)&s->d_count, nb * 4));
    if (rank) PG_HIP(hipMalloc((void**)&s->d_rank, std::max<size_t>(nb * k, c->rank_item_cap) * 4));
    if (c->e) {
        PG_HIP(hipMalloc((void**)&s->d_fused, nb * k * 8));
        PG_HIP(hipMalloc((void**)&s->d_order, nb * k * 4));
        PG_HIP(hipMalloc((void**)&s->d_page, page_bytes(c)));
    }
    return pipe_run_acquire(s->ctx, &s->run);
}

void free_slot(pg_coalescer* c, Slot* s) {
    if (s->run) pipe_run_release(s->ctx, s->run);
    if (s->done) hipEventDestroy(s->done);
    if (s->computed) hipEventDestroy(s->computed);
    for (void* p : {(void*)s->h_vec, (void*)s->h_cand, (void*)s->h_off, (void*)s->h_out})
        if (p) hipHostFree(p);
    for (void* p : {(void*)s->d_vec, (void*)s->d_cand, (void*)s->d_off, (void*)s->d_rows, (void*)s->d_recall, (void*)s->d_rank,
                    (void*)s->d_fused, (void*)s->d_order, (void*)s->d_count, (void*)s->d_page})
        if (p) hipFree(p);
    delete s;
}

// Outputs device → pinned host on the copy stream (so the next batch's kernels do not queue behind a PCIe transfer),
// then the completion event.  Called behind the batch's kernels, and again when the verification patched single
// requests in place.
int slot_copy_out(pg_coalescer* c, Slot* s) {
    hipStream_t st = s->ctx->stream;
    const uint32_t nq = s->n_req;
    if (s->kind == kRank) {
        PG_HIP(hipEventRecord(s->computed, st));
        PG_HIP(hipStreamWaitEvent(c->copy_stream, s->computed, 0));
        PG_HIP(hipMemcpyAsync(s->h_out, s->d_rank, (size_t)s->n_items * 4, hipMemcpyDeviceToHost, c->copy_stream));
        PG_HIP(hipEventRecord(s->done, c->copy_stream));
        return PG_OK;
    }
    if (s->kind == kRecall) {
        PG_HIP(hipEventRecord(s->computed, st));
        PG_HIP(hipStreamWaitEvent(c->copy_stream, s->computed, 0));
        const size_t nk = (size_t)nq * c->k;
        PG_HIP(hipMemcpyAsync(s->h_out, s->d_rows, nk * 8, hipMemcpyDeviceToHost, c->copy_stream));
        PG_HIP(hipMemcpyAsync(s->h_out + (size_t)c->max_batch * c->k * 8, s->d_recall, nk * 4, hipMemcpyDeviceToHost, c->copy_stream));
        PG_HIP(hipEventRecord(s->done, c->copy_stream));
        return PG_OK;
    }
    // recommend: the largest page any request of the batch asked for
    const uint32_t top = s->n_items;
    const size_t np = (size_t)nq * top;
    uint64_t* p_rows = (uint64_t*)s->d_page;
    double* p_fused = (double*)(p_rows + np);
    float* p_recall = (float*)(p_fused + np);
    float* p_rank = p_recall + np;
    page_gather_kernel<<<(uint32_t)((np + 255) / 256), 256, 0, st>>>(s->d_order, s->d_rows, s->d_recall, s->d_rank, s->d_fused, nq,
                                                                    c->k, top, p_rows, p_fused, p_recall, p_rank);
    PG_HIP(hipGetLastError());
    PG_HIP(hipEventRecord(s->computed, st));
    PG_HIP(hipStreamWaitEvent(c->copy_stream, s->computed, 0));
    PG_HIP(hipMemcpyAsync(s->h_out, s->d_page, np * 24, hipMemcpyDeviceToHost, c->copy_stream));
    PG_HIP(hipMemcpyAsync(s->h_out + page_bytes(c), s->d_count, (size_t)nq * 4, hipMemcpyDeviceToHost, c->copy_stream));
    PG_HIP(hipEventRecord(s->done, c->copy_stream));
    return PG_OK;
}

// Enqueue the slot's batch: inputs host → device, the kernels, the copy-out.  first = false: the recall plan of a
// recall / recommend batch did not hold; run its next plan and everything behind it again.
int slot_enqueue(pg_coalescer* c, Slot* s, bool first) {
    pg_ctx* ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t nq = (uint32_t)s->n_req;
    int rc;
    if (s->kind == kRank) {
        PG_HIP(hipMemcpyAsync(s->d_vec, s->h_vec, (size_t)nq * c->d_user * 4, hipMemcpyHostToDevice, st));
        PG_HIP(hipMemcpyAsync(s->d_cand, s->h_cand, (size_t)s->n_items * 4, hipMemcpyHostToDevice, st));
        PG_HIP(hipMemcpyAsync(s->d_off, s->h_off, ((size_t)nq + 1) * 4, hipMemcpyHostToDevice, st));
        {
            std::lock_guard<std::mutex> g(ctx->mu);
            if ((rc = rank_dnn3_dev_locked(ctx, c->m, c->t, s->d_vec, s->d_cand, s->d_off, nq, s->n_items, s->d_rank))) return rc;
        }
        return slot_copy_out(c, s);
    }
    if (first) PG_HIP(hipMemcpyAsync(s->d_vec, s->h_vec, (size_t)nq * c->dim * 4, hipMemcpyHostToDevice, st));
    if (s->kind == kRecall) {
        {
            std::lock_guard<std::mutex> g(ctx->mu);
            RecallJob& j = s->run->job;
            if (first) {
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
                if ((rc = recall_job_prepare(&j))) return rc;
            }
            s->run->patched = false;
            if ((rc = recall_job_enqueue(&j))) return rc;
        }
        return slot_copy_out(c, s);
    }
    // recommend
    RecommendCall& rc_call = s->call;
    rc_call = RecommendCall();
    rc_call.t = c->t;
    rc_call.m = c->m;
    rc_call.e = c->e;
    rc_call.var_src = c->var_src.data();
    rc_call.nv = (int)c->var_src.size();
    rc_call.d_queries = s->d_vec;
    rc_call.nq = nq;
    rc_call.k = c->k;
    rc_call.d_rows = s->d_rows;
    rc_call.d_recall = s->d_recall;
    rc_call.d_rank = s->d_rank;
    rc_call.d_fused = s->d_fused;
    rc_call.d_order = s->d_order;
    rc_call.d_count = s->d_count;
    if ((rc = recommend_enqueue(ctx, rc_call, s->run, first))) return rc;
    uint32_t top = 1;
    for (const Req* r : s->reqs) top = std::max(top, r->n);
    s->n_items = top;
    return slot_copy_out(c, s);
}

// finish every request of a slot with (rc, message of this thread) and wake the callers
void slot_fail(Slot* s, int rc) {
    const char* msg = pg_last_error();
    for (Req* r : s->reqs) fail_req(r, rc, msg);
}

void slot_wake(pg_coalescer* c, Slot* s) {
    // callers release the slot: the last one to have copied its slice returns it to the free list
    s->pending.store((uint32_t)s->reqs.size(), std::memory_order_release);
    std::vector<Req*> reqs;
    reqs.swap(s->reqs);                    // a woken caller's Req lives on its stack: do not touch it after the wake
    for (Req* r : reqs) {
        r->done.store(1, std::memory_order_release);
        futex_wake_all(&r->done);
    }
}

void dispatcher_main(pg_coalescer* c) {
    hipSetDevice(c->ctx->device);
    std::unique_lock<std::mutex> lk(c->mu);
    while (true) {
        if (c->stop) break;
        // Which flavour goes next?  A batch is ready when it is full, or when nothing is in flight on the device and
        // its oldest request has waited max_wait_us; among ready flavours the one whose head is oldest wins.
        const bool idle = c->inflight.empty();
        const auto now = Clock::now();
        int kind = -1;
        bool any = false;
        auto earliest = Clock::time_point::max();
        for (int f = 0; f < 3; ++f) {
            std::deque<Req*>& qf = c->queue[f];
            if (qf.empty()) continue;
            any = true;
            bool full;
            if (f == kRank) {
                size_t items = 0;
                for (const Req* r : qf) items += r->n;
                full = qf.size() >= c->max_rank_reqs || items >= (size_t)c->max_batch * c->k;
            } else {
                full = qf.size() >= c->max_batch;
            }
            const auto deadline = qf.front()->arrived + std::chrono::microseconds(c->max_wait_us);
            if (deadline < earliest) earliest = deadline;
            if ((full || (idle && now >= deadline)) && (kind < 0 || qf.front()->arrived < c->queue[kind].front()->arrived)) kind = f;
        }
        if (kind < 0 || c->free_slots.empty()) {
            // nothing ready (or `depth` batches already in flight: keep collecting).  A new request, a completion
            // or a released slot wakes us; an idle device additionally at the oldest request's deadline.
            if (any && idle && kind < 0) c->cv_dispatch.wait_until(lk, earliest);
            else c->cv_dispatch.wait(lk);
            continue;
        }
        std::deque<Req*>& q = c->queue[kind];
        Slot* s = c->free_slots.front();               // oldest first: consecutive batches alternate between the contexts
        c->free_slots.erase(c->free_slots.begin());
        s->kind = (Flavour)kind;
        s->reqs.clear();
        s->n_items = 0;
        if (kind == kRank) {
            while (!q.empty() && s->reqs.size() < c->max_rank_reqs && s->n_items + q.front()->n <= c->rank_item_cap) {
                Req* r = q.front();
                q.pop_front();
                r->item0 = s->n_items;
                s->n_items += r->n;
                s->reqs.push_back(r);
            }
        } else {
            while (!q.empty() && s->reqs.size() < c->max_batch) {
                s->reqs.push_back(q.front());
                q.pop_front();
            }
        }
        lk.unlock();
        // stage the inputs (the callers are blocked: their buffers are stable)
        const uint32_t nq = (uint32_t)s->reqs.size();
        s->n_req = nq;
        const uint32_t w = kind == kRank ? c->d_user : c->dim;
        for (uint32_t i = 0; i < nq; ++i) {
            Req* r = s->reqs[i];
            r->slot = s;
            r->index = i;
            memcpy(s->h_vec + (size_t)i * w, r->vec, (size_t)w * 4);
            if (kind == kRank) {
                s->h_off[i] = r->item0;
                memcpy(s->h_cand + r->item0, r->cand, (size_t)r->n * 4);
            }
        }
        if (kind == kRank) s->h_off[nq] = s->n_items;
        s->enqueued = Clock::now();
        const int rc = slot_enqueue(c, s, true);
        lk.lock();
        c->stats.requests[kind] += nq;
        c->stats.batches[kind] += 1;
        c->stats.largest_batch[kind] = std::max<uint64_t>(c->stats.largest_batch[kind], nq);
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
            r->done.store(1, std::memory_order_release);
            futex_wake_all(&r->done);
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
        bool replanned = false;
        for (;;) {
            if (hipEventSynchronize(s->done) != hipSuccess) {
                set_error("pg_coalescer: %s", hipGetErrorString(hipGetLastError()));
                rc = PG_ERR_DEVICE;
                break;
            }
            if (s->kind == kRank) break;
            bool ok = false;
            // (recall_job_check + finish under ctx->mu; a few failed requests are re-run in place)
            if ((rc = recommend_verify(s->ctx, s->run, &ok, s->kind == kRecommend ? &s->call : nullptr))) break;
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
        if (rc) {
            slot_fail(s, rc);
        } else if (s->kind == kRecommend) {
            for (Req* r : s->reqs)
                if (s->run->h_status[kExprFlagAt + r->index]) {
                    set_expr_arith_error(c->e);
                    fail_req(r, PG_ERR_ARITH, pg_last_error());
                }
        }
        const double ms = std::chrono::duration<double, std::milli>(Clock::now() - s->enqueued).count();
        const int kind = s->kind;
        lk.lock();
        c->inflight.pop_front();
        c->stats.device_ms[kind] += ms;
        if (replanned) c->stats.replans++;
        lk.unlock();
        slot_wake(c, s);
        lk.lock();
        c->cv_dispatch.notify_one();               // the device may be idle now: a waiting partial batch can go
    }
}

// caller side: queue the request, sleep until a worker finished it, copy the slice, release the slot
int submit_and_wait(pg_coalescer* c, Req* r) {
    r->arrived = Clock::now();
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (c->stop) {
            set_error("pg_coalescer: already shut down");
            return PG_ERR_INVALID;
        }
        c->queue[r->kind].push_back(r);
    }
    c->cv_dispatch.notify_one();
    while (r->done.load(std::memory_order_acquire) == 0) futex_wait(&r->done, 0);
    return PG_OK;
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

}  // namespace
}  // namespace pg

extern "C" {

int pg_coalescer_create(pg_ctx* ctx, const pg_table* t, const pg_model* m, const pg_expr* e, const char* rank_var,
                        const pg_coalescer_config* cfg, pg_coalescer** out) {
    PG_REQUIRE(ctx && t && cfg && out, "pg_coalescer_create: NULL argument");
    PG_REQUIRE(cfg->k >= 1 && cfg->k <= 16384, "pg_coalescer_create: k=%u unsupported (1..16384)", cfg->k);
    const uint32_t max_q = t->dim <= 128 ? (uint32_t)pg::kMaxQueries : 32u;
    PG_REQUIRE(cfg->max_batch <= max_q, "pg_coalescer_create: max_batch %u exceeds %u queries per pass at dim %u",
               cfg->max_batch, max_q, t->dim);
    PG_REQUIRE(cfg->depth <= 4, "pg_coalescer_create: depth %u (at most 4)", cfg->depth);
    PG_REQUIRE(cfg->max_top_n <= cfg->k, "pg_coalescer_create: max_top_n %u exceeds k %u", cfg->max_top_n, cfg->k);
    PG_REQUIRE(!e || (m && rank_var), "pg_coalescer_create: a RankScore expression needs a model and its name");
    if (m) {
        PG_REQUIRE(m->kind == PG_MODEL_DNN3 && m->d_item == t->dim, "pg_coalescer_create: the model must be DNN3 over the table's rows");
        PG_REQUIRE(!e || m->d_user == t->dim, "pg_coalescer_create: recommend needs d_user = the table's dim (the user vector is the query)");
    }
    PG_HIP(hipSetDevice(ctx->device));
    pg_coalescer* c = new pg_coalescer();
    c->ctx = ctx;
    c->t = t;
    c->m = m;
    c->e = e;
    int rc;
    if (e && (rc = pg::recommend_bind_vars(e, rank_var, &c->var_src, "pg_coalescer_create"))) {
        delete c;
        return rc;
    }
    c->k = cfg->k;
    c->max_batch = cfg->max_batch ? cfg->max_batch : max_q;
    c->max_wait_us = cfg->max_wait_us ? cfg->max_wait_us : 100;
    c->depth = cfg->depth ? cfg->depth : 2;
    c->max_top_n = cfg->max_top_n ? cfg->max_top_n : cfg->k;
    c->max_rank_items = cfg->max_rank_items ? cfg->max_rank_items : cfg->k;
    c->dim = t->dim;
    c->d_user = m ? m->d_user : 0;
    // a rank batch: as many candidates as a full recommend batch ranks, from at most 16384 calls
    c->rank_item_cap = std::max<uint32_t>(c->max_batch * c->k, c->max_rank_items);
    c->max_rank_reqs = 16384;
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        pg::set_error("pg_coalescer_create: %s", hipGetErrorString(hipGetLastError()));
        delete c;
        return PG_ERR_DEVICE;
    }
    if (c->depth >= 2 && pg_init(ctx->device, nullptr, &c->sibling) != PG_OK) c->sibling = nullptr;   // (optional: one stream works too)
    for (uint32_t i = 0; i < c->depth; ++i) {
        pg::Slot* s = new pg::Slot();
        s->id = (int)i;
        s->ctx = (c->sibling && (i & 1)) ? c->sibling : ctx;
        if ((rc = pg::alloc_slot(c, s))) {
            pg::free_slot(c, s);
            for (pg::Slot* o : c->slots) pg::free_slot(c, o);
            hipStreamDestroy(c->copy_stream);
            if (c->sibling) pg_shutdown(c->sibling);
            delete c;
            return rc;
        }
        c->slots.push_back(s);
        c->free_slots.push_back(s);
    }
    // build the table's shadow now rather than inside the first batch
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        if ((rc = pg::ensure_table_stats(ctx, t))) {
            for (pg::Slot* o : c->slots) pg::free_slot(c, o);
            hipStreamDestroy(c->copy_stream);
            if (c->sibling) pg_shutdown(c->sibling);
            delete c;
            return rc;
        }
    }
    c->dispatcher = std::thread(pg::dispatcher_main, c);
    c->completer = std::thread(pg::completer_main, c);
    *out = c;
    return PG_OK;
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
    hipStreamSynchronize(c->copy_stream);
    hipStreamSynchronize(c->ctx->stream);
    if (c->sibling) hipStreamSynchronize(c->sibling->stream);
    for (pg::Slot* s : c->slots) pg::free_slot(c, s);
    hipStreamDestroy(c->copy_stream);
    if (c->sibling) pg_shutdown(c->sibling);
    delete c;
    return PG_OK;
}

int pg_coalescer_recall(pg_coalescer* c, const float* query, uint64_t* out_rows, float* out_scores,
                        uint32_t* out_count) {
    PG_REQUIRE(c && query && out_rows && out_scores, "pg_coalescer_recall: NULL argument");
    pg::Req r;
    r.kind = pg::kRecall;
    r.vec = query;
    int rc;
    if ((rc = pg::submit_and_wait(c, &r))) return rc;
    pg::Slot* s = r.slot;
    if (r.rc == PG_OK) {
        const size_t k = c->k;
        memcpy(out_rows, s->h_out + (size_t)r.index * k * 8, k * 8);
        memcpy(out_scores, s->h_out + (size_t)c->max_batch * k * 8 + (size_t)r.index * k * 4, k * 4);
        if (out_count) *out_count = s->run->h_status[1 + r.index];
    } else {
        pg::set_error("%s", r.err);
    }
    if (s) pg::release_slot(c, s);
    return r.rc;
}

int pg_coalescer_rank_dnn3(pg_coalescer* c, const float* user_vec, const uint32_t* cand_rows, uint32_t n,
                           float* out_scores) {
    PG_REQUIRE(c && user_vec, "pg_coalescer_rank_dnn3: NULL argument");
    PG_REQUIRE(c->m, "pg_coalescer_rank_dnn3: the coalescer was created without a model");
    PG_REQUIRE(n <= c->max_rank_items, "pg_coalescer_rank_dnn3: %u candidates exceed max_rank_items %u", n, c->max_rank_items);
    if (n == 0) return PG_OK;
    PG_REQUIRE(cand_rows && out_scores, "pg_coalescer_rank_dnn3: NULL argument");
    for (uint32_t i = 0; i < n; ++i)
        PG_REQUIRE(cand_rows[i] < c->t->rows, "pg_coalescer_rank_dnn3: candidate %u row %u outside table of %llu rows", i,
                   cand_rows[i], (unsigned long long)c->t->rows);
    pg::Req r;
    r.kind = pg::kRank;
    r.vec = user_vec;
    r.cand = cand_rows;
    r.n = n;
    int rc;
    if ((rc = pg::submit_and_wait(c, &r))) return rc;
    pg::Slot* s = r.slot;
    if (r.rc == PG_OK) memcpy(out_scores, s->h_out + (size_t)r.item0 * 4, (size_t)n * 4);
    else pg::set_error("%s", r.err);
    if (s) pg::release_slot(c, s);
    return r.rc;
}

int pg_coalescer_recommend(pg_coalescer* c, const float* user_vec, uint32_t top_n, uint64_t* out_rows,
                           float* out_recall_scores, float* out_rank_scores, double* out_fused,
                           uint32_t* out_count) {
    PG_REQUIRE(c && user_vec && out_rows && out_recall_scores && out_rank_scores && out_fused,
               "pg_coalescer_recommend: NULL argument");
    PG_REQUIRE(c->e, "pg_coalescer_recommend: the coalescer was created without a RankScore expression");
    PG_REQUIRE(top_n >= 1 && top_n <= c->max_top_n, "pg_coalescer_recommend: top_n %u outside 1..%u", top_n, c->max_top_n);
    pg::Req r;
    r.kind = pg::kRecommend;
    r.vec = user_vec;
    r.n = top_n;
    int rc;
    if ((rc = pg::submit_and_wait(c, &r))) return rc;
    pg::Slot* s = r.slot;
    if (r.rc == PG_OK) {
        const size_t nq_top = (size_t)s->n_items;                       // page width of the batch's image: planes are [n_req][nq_top]
        const uint32_t* counts = (const uint32_t*)(s->h_out + pg::page_bytes(c));
        const uint32_t batch = s->n_req;
        const size_t np = (size_t)batch * nq_top;
        const uint64_t* p_rows = (const uint64_t*)s->h_out;
        const double* p_fused = (const double*)(p_rows + np);
        const float* p_recall = (const float*)(p_fused + np);
        const float* p_rank = p_recall + np;
        const size_t o = (size_t)r.index * nq_top;
        memcpy(out_rows, p_rows + o, (size_t)top_n * 8);
        memcpy(out_fused, p_fused + o, (size_t)top_n * 8);
        memcpy(out_recall_scores, p_recall + o, (size_t)top_n * 4);
        memcpy(out_rank_scores, p_rank + o, (size_t)top_n * 4);
        if (out_count) *out_count = std::min(top_n, counts[r.index]);
    } else {
        pg::set_error("%s", r.err);
    }
    if (s) pg::release_slot(c, s);
    return r.rc;
}

int pg_coalescer_stats(pg_coalescer* c, pg_coalescer_stats_t* out) {
    PG_REQUIRE(c && out, "pg_coalescer_stats: NULL argument");
    std::lock_guard<std::mutex> g(c->mu);
    *out = c->stats;
    return PG_OK;
}

}  // extern "C"
