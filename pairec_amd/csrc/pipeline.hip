// pipeline.hip — the whole hot path for a batch of requests as ONE stream of device work.
//
// service/user_recommend.go:83-151 restricted to the path: VectorRecall.GetCandidateItems
// (service/recall/vector_recall.go:32-123) → RankService.Rank with one DNN3 algorithm
// (service/rank/rank_service.go:102-372) → RankScore fusion (utils/ast/ast.go:215-268) → ItemRankScoreSort
// (sort/item_rank_score.go:26-32).  The stages are enqueued back to back; the two facts the host has to know —
// did the recall's plan hold, did a RankScore divide by zero — travel to pinned memory behind the last kernel and
// are read once, at the end (round 1 synchronised inside the recall and inside the fusion: ~0.2 ms of idle GPU per
// batch, and no way to queue a second batch behind the first).
#include "pipeline.hpp"

#include <algorithm>

namespace pg {

// vars[v][i] = (double) (src[v] ? rank[i] : recall[i]) — the float32 → float64 widening every response decoder of
// the reference performs (algorithm/eas/easyrec_response.go:479-483), for all variables of the expression at once
__global__ void bind_vars_kernel(const float* __restrict__ recall, const float* __restrict__ rank, uint32_t n,
                                 uint32_t nv, uint32_t src_mask, double* __restrict__ vars) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = (double)recall[i], b = (double)rank[i];
    for (uint32_t v = 0; v < nv; ++v) vars[(size_t)v * n + i] = ((src_mask >> v) & 1u) ? b : a;
}

// A table with fewer than k rows leaves padding slots (row = UINT64_MAX, recall score = -inf) at the end of every
// request.  They are not items: their model score is reported as 0 and their fused score as NaN, which the sort
// places last in either direction, so a page never starts with them.
__global__ void mask_pads_kernel(const uint64_t* __restrict__ rows, uint32_t n, float* __restrict__ rank,
                                 double* __restrict__ fused) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || rows[i] != ~0ull) return;
    rank[i] = 0.0f;
    fused[i] = __longlong_as_double(0x7FF8000000000000ll);
}

int pipe_run_acquire(pg_ctx* ctx, PipeRun** out) {
    {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        if (!ctx->pipe_free.empty()) {
            *out = ctx->pipe_free.back();
            ctx->pipe_free.pop_back();
            return PG_OK;
        }
    }
    PG_HIP(hipSetDevice(ctx->device));
    PipeRun* r = new PipeRun();
    if (hipHostMalloc((void**)&r->h_status, kPipeStatusWords * 4) != hipSuccess || hipEventCreate(&r->done) != hipSuccess) {
        set_error("pipeline: cannot allocate a batch's status block: %s", hipGetErrorString(hipGetLastError()));
        if (r->h_status) hipHostFree(r->h_status);
        delete r;
        return PG_ERR_DEVICE;
    }
    *out = r;
    return PG_OK;
}

void pipe_run_release(pg_ctx* ctx, PipeRun* r) {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    ctx->pipe_free.push_back(r);
}

void pipe_pool_destroy(pg_ctx* ctx) {
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    for (PipeRun* r : ctx->pipe_free) {
        for (hipEvent_t e : r->events) hipEventDestroy(e);
        if (r->done) hipEventDestroy(r->done);
        if (r->h_status) hipHostFree(r->h_status);
        delete r;
    }
    ctx->pipe_free.clear();
}

int recommend_bind_vars(const pg_expr* e, const char* rank_var, std::vector<int>* var_src, const char* who) {
    const int nv = pg_expr_num_vars(e);
    if (nv > 32) {
        set_error("%s: RankScore has %d variables (at most 32)", who, nv);
        return PG_ERR_UNSUPPORTED;
    }
    var_src->assign((size_t)nv, 0);
    for (int i = 0; i < nv; ++i) {
        const char* name = pg_expr_var_name(e, i);
        if (!strcmp(name, rank_var)) (*var_src)[(size_t)i] = 1;
        else if (!strcmp(name, "current_score")) (*var_src)[(size_t)i] = 0;
        else {
            set_error("%s: RankScore variable \"%s\" is neither \"%s\" nor current_score", who, name, rank_var);
            return PG_ERR_INVALID;
        }
    }
    return PG_OK;
}

// the stages behind the recall for requests [q0, q0 + nq) of the call (caller holds ctx->mu; d_err is indexed from 0)
static int recommend_post_locked(pg_ctx* ctx, const RecommendCall& c, uint32_t q0, uint32_t nq, uint32_t* d_local,
                                 uint32_t* d_off, uint32_t* d_err, double* d_vars) {
    const uint32_t n = nq * c.k;
    const size_t o = (size_t)q0 * c.k;
    int rc;
    if ((rc = uniform_offsets_locked(ctx, nq, c.k, d_off))) return rc;
    if ((rc = rows_to_local_locked(ctx, c.t, c.d_rows + o, n, d_local, nullptr))) return rc;
    if ((rc = rank_dnn3_dev_locked(ctx, c.m, c.t, c.d_queries + (size_t)q0 * c.t->dim, d_local, d_off, nq, n, c.d_rank + o))) return rc;
    uint32_t mask = 0;
    for (int i = 0; i < c.nv; ++i) mask |= (c.var_src[i] ? 1u : 0u) << i;
    if (c.nv > 0) {
        bind_vars_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(c.d_recall + o, c.d_rank + o, n, (uint32_t)c.nv, mask, d_vars);
        PG_HIP(hipGetLastError());
    }
    PG_HIP(hipMemsetAsync(d_err, 0, (size_t)kMaxQueries * 4, ctx->stream));
    if ((rc = expr_eval_enqueue_locked(ctx, c.e, d_vars, n, c.d_fused + o, d_err, c.k))) return rc;
    if (c.t->rows < c.k) {
        mask_pads_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(c.d_rows + o, n, c.d_rank + o, c.d_fused + o);
        PG_HIP(hipGetLastError());
    }
    return sort_dev_locked(ctx, c.d_fused + o, d_off, nq, n, c.k, 1, c.d_order + o);
}

struct PostScratch {
    uint32_t *d_local, *d_off, *d_err;
    double* d_vars;
};
static int post_scratch(pg_ctx* ctx, const RecommendCall& c, uint32_t nq, PostScratch* ps) {
    const uint32_t n = nq * c.k;
    void* buf;
    int rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_local = al((size_t)n * 4), b_off = al((size_t)(nq + 1) * 4), b_err = al((size_t)kMaxQueries * 4);
    const size_t b_vars = al((size_t)std::max(c.nv, 1) * n * 8);
    if ((rc = scratch_reserve(ctx, 8, b_local + b_off + b_err + b_vars, &buf))) return rc;
    ps->d_local = (uint32_t*)buf;
    ps->d_off = (uint32_t*)((char*)buf + b_local);
    ps->d_err = (uint32_t*)((char*)buf + b_local + b_off);
    ps->d_vars = (double*)((char*)buf + b_local + b_off + b_err);
    return PG_OK;
}

int recommend_enqueue(pg_ctx* ctx, const RecommendCall& c, PipeRun* r, bool first) {
    std::lock_guard<std::mutex> g(ctx->mu);
    int rc;
    PostScratch ps;
    if ((rc = post_scratch(ctx, c, c.nq, &ps))) return rc;
    r->patched = false;
    if (first) {
        RecallJob& j = r->job;
        j = RecallJob();
        j.ctx = ctx;
        j.t = c.t;
        j.d_queries = c.d_queries;
        j.nq = c.nq;
        j.k = c.k;
        j.d_out_rows = c.d_rows;
        j.d_out_scores = c.d_recall;
        j.d_out_count = c.d_count;
        j.h_status = r->h_status;
        j.events = &r->events;
        if ((rc = recall_job_prepare(&j))) return rc;
    }
    if ((rc = recall_job_enqueue(&r->job))) return rc;
    if ((rc = recommend_post_locked(ctx, c, 0, c.nq, ps.d_local, ps.d_off, ps.d_err, ps.d_vars))) return rc;
    PG_HIP(hipMemcpyAsync(r->h_status + kExprFlagAt, ps.d_err, (size_t)c.nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipEventRecord(r->done, ctx->stream));
    return PG_OK;
}

int recommend_verify(pg_ctx* ctx, PipeRun* r, bool* ok, const RecommendCall* c) {
    std::lock_guard<std::mutex> g(ctx->mu);
    int rc;
    if ((rc = recall_job_check(&r->job, ok))) return rc;
    RecallJob& j = r->job;
    if (!*ok && !j.failed.empty() && j.failed.size() <= kMaxPatchQueries && j.nq > 1) {
        // The pilot's threshold was too high for a few requests only (a 1e-4 event per request at the default margin):
        // re-run those requests — recall from the growing-chunk plan, then the stages behind it — synchronously and in
        // place, instead of the whole batch.  The nested recalls use the context's own status block.
        const std::vector<uint32_t> failed = j.failed;
        uint32_t counts[kMaxQueries];
        for (uint32_t q = 0; q < j.nq; ++q) counts[q] = r->h_status[1 + q];
        if ((rc = recall_patch_failed_locked(&j, counts))) return rc;
        for (uint32_t q = 0; q < j.nq; ++q) r->h_status[1 + q] = counts[q];
        if (c) {
            PostScratch ps;
            if ((rc = post_scratch(ctx, *c, 1, &ps))) return rc;
            for (uint32_t q : failed) {
                if ((rc = recommend_post_locked(ctx, *c, q, 1, ps.d_local, ps.d_off, ps.d_err, ps.d_vars))) return rc;
                PG_HIP(hipMemcpyAsync(r->h_status + kExprFlagAt + q, ps.d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
                PG_HIP(hipStreamSynchronize(ctx->stream));
            }
        }
        PG_HIP(hipStreamSynchronize(ctx->stream));
        r->patched = true;
        *ok = true;
    }
    if (*ok) recall_job_finish(&r->job);
    return PG_OK;
}

}  // namespace pg

struct pg_ticket {
    pg::RecommendCall call;
    std::vector<int> var_src;
    pg::PipeRun* run = nullptr;
};

extern "C" {

int pg_recommend_dnn3_begin(pg_ctx* ctx, const pg_table* t, const pg_model* m, const pg_expr* e, const char* rank_var,
                            const float* d_queries, uint32_t nq, uint32_t k, uint64_t* d_out_rows,
                            float* d_out_recall_scores, float* d_out_rank_scores, double* d_out_fused,
                            uint32_t* d_out_order, uint32_t* d_out_count, pg_ticket** out) {
    PG_REQUIRE(ctx && t && m && e && rank_var && d_queries && d_out_rows && d_out_recall_scores && d_out_rank_scores &&
                   d_out_fused && d_out_order && out,
               "pg_recommend_dnn3: NULL argument");
    PG_REQUIRE(nq > 0 && nq <= (uint32_t)pg::kMaxQueries && k > 0 && k <= 16384, "pg_recommend_dnn3: bad nq / k");
    PG_REQUIRE(m->kind == PG_MODEL_DNN3 && t->dim == m->d_item && m->d_user == t->dim,
               "pg_recommend_dnn3: the model must be DNN3 with d_user = d_item = the table's dim");
    pg_ticket* tk = new pg_ticket();
    int rc;
    if ((rc = pg::recommend_bind_vars(e, rank_var, &tk->var_src, "pg_recommend_dnn3"))) {
        delete tk;
        return rc;
    }
    pg::RecommendCall& c = tk->call;
    c.t = t; c.m = m; c.e = e; c.var_src = tk->var_src.data(); c.nv = (int)tk->var_src.size();
    c.d_queries = d_queries; c.nq = nq; c.k = k;
    c.d_rows = d_out_rows; c.d_recall = d_out_recall_scores; c.d_rank = d_out_rank_scores;
    c.d_fused = d_out_fused; c.d_order = d_out_order; c.d_count = d_out_count;
    if ((rc = pg::pipe_run_acquire(ctx, &tk->run)) || (rc = pg::recommend_enqueue(ctx, c, tk->run, true))) {
        if (tk->run) pg::pipe_run_release(ctx, tk->run);
        delete tk;
        return rc;
    }
    *out = tk;
    return PG_OK;
}

int pg_recommend_end(pg_ctx* ctx, pg_ticket* tk, double* scan_ms) {
    PG_REQUIRE(ctx && tk, "pg_recommend_end: NULL argument");
    int rc = PG_OK;
    for (bool ok = false; !ok;) {
        if (hipEventSynchronize(tk->run->done) != hipSuccess) {
            pg::set_error("pg_recommend_end: %s", hipGetErrorString(hipGetLastError()));
            rc = PG_ERR_DEVICE;
            break;
        }
        if ((rc = pg::recommend_verify(ctx, tk->run, &ok, &tk->call))) break;
        if (!ok && (rc = pg::recommend_enqueue(ctx, tk->call, tk->run, false))) break;
    }
    if (!rc) {
        if (scan_ms) *scan_ms = tk->run->job.scan_ms;
        for (uint32_t q = 0; q < tk->call.nq; ++q)
            if (tk->run->h_status[pg::kExprFlagAt + q]) {
                pg::set_expr_arith_error(tk->call.e);
                rc = PG_ERR_ARITH;
                break;
            }
    }
    pg::pipe_run_release(ctx, tk->run);
    delete tk;
    return rc;
}

int pg_recommend_dnn3_dev(pg_ctx* ctx, const pg_table* t, const pg_model* m, const pg_expr* e, const char* rank_var,
                          const float* d_queries, uint32_t nq, uint32_t k, uint64_t* d_out_rows,
                          float* d_out_recall_scores, float* d_out_rank_scores, double* d_out_fused,
                          uint32_t* d_out_order, uint32_t* d_out_count) {
    pg_ticket* tk = nullptr;
    int rc;
    if ((rc = pg_recommend_dnn3_begin(ctx, t, m, e, rank_var, d_queries, nq, k, d_out_rows, d_out_recall_scores,
                                      d_out_rank_scores, d_out_fused, d_out_order, d_out_count, &tk)))
        return rc;
    return pg_recommend_end(ctx, tk, nullptr);
}

}  // extern "C"
