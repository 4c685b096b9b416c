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
#include <chrono>
#include <thread>

namespace pg {

// vars[v][i] = (double) (src[v] >= 0 ? rank plane src[v] : recall)[i] — the float32 → float64 widening every response
// decoder of the reference performs (algorithm/eas/easyrec_response.go:479-483), for all variables of the expression
struct VarSrc { int8_t src[32]; };
// (src >= 0: score plane; -1: Item.Score; <= -2: the f64 result of score rewrite -2 - src, rw [n_rewrites][n])
__global__ void bind_vars_kernel(const float* __restrict__ recall, const float* __restrict__ rank, size_t rank_stride,
                                 uint32_t n, uint32_t nv, VarSrc vs, double* __restrict__ vars, uint32_t* __restrict__ err,
                                 const double* __restrict__ rw, int zero_err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_err && i < (uint32_t)kMaxQueries) err[i] = 0; // the RankScore flags of the evaluations behind this launch
    if (i >= n) return;
    const double a = (double)recall[i];
    for (uint32_t v = 0; v < nv; ++v) {
        const int sa = vs.src[v];
        vars[(size_t)v * n + i] = sa >= 0 ? (double)rank[(size_t)sa * rank_stride + i] : (sa == -1 ? a : rw[(size_t)(-2 - sa) * n + i]);
    }
}

// A table with fewer than k rows leaves padding slots (row = UINT64_MAX, recall score = -inf) at the end of every
// request.  They are not items: their model scores are reported as 0 and their fused score as NaN, which the sort
// places last in either direction, so a page never starts with them.
__global__ void mask_pads_kernel(const uint64_t* __restrict__ rows, uint32_t n, float* __restrict__ rank, size_t rank_stride,
                                 int n_algos, double* __restrict__ fused) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || rows[i] != ~0ull) return;
    for (int a = 0; a < n_algos; ++a) rank[(size_t)a * rank_stride + i] = 0.0f;
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

// var_src: the RankScore's variables first, then those of every score rewrite in order (RecommendCall::nv counts the
// RankScore's only).  A RankScore variable that names a rewrite's source reads the rewritten score (AddAlgoScores overwrites
// the algorithm's, module/item.go:177-188); a rewrite's own variables read the scores as the algorithms left them — the
// reference fills a map from the un-rewritten item first (rank_service.go:343-353).
int recommend_bind_vars(const pg_expr* e, const char* const* names, int n_algos, std::vector<int>* var_src, const char* who) {
    const int nv = pg_expr_num_vars(e), n_rw = expr_num_rewrites(e);
    int worst = nv;
    for (int r = 0; r < n_rw; ++r) worst = std::max(worst, expr_rewrite_num_vars(e, r));
    if (worst > 32) {
        set_error("%s: RankScore (or one of its score rewrites) has %d variables (at most 32)", who, worst);
        return PG_ERR_UNSUPPORTED;
    }
    var_src->clear();
    auto plane_of = [&](const char* name) {
        int found = -2;
        for (int a = 0; a < n_algos; ++a)
            if (names[a] && !strcmp(name, names[a])) found = a;
        if (found == -2 && !strcmp(name, "current_score")) found = -1;
        return found;                                     // -2: unknown
    };
    for (int i = 0; i < nv; ++i) {
        const char* name = pg_expr_var_name(e, i);
        int found = -2;
        bool rewritten = false;                           // (-2 - r is also the "unknown" mark for r = 0)
        for (int r = 0; r < n_rw; ++r)
            if (!strcmp(name, expr_rewrite_source(e, r))) {
                found = -2 - r;
                rewritten = true;
            }
        if (!rewritten && (found = plane_of(name)) == -2) {
            set_error("%s: RankScore variable \"%s\" is neither a rank algorithm of the scene, a ScoreRewrite source nor current_score", who, name);
            return PG_ERR_INVALID;
        }
        var_src->push_back(found);
    }
    for (int r = 0; r < n_rw; ++r)
        for (int i = 0; i < expr_rewrite_num_vars(e, r); ++i) {
            const char* name = expr_rewrite_var_name(e, r, i);
            const int found = plane_of(name);
            if (found == -2) {
                set_error("%s: ScoreRewrite[\"%s\"] variable \"%s\" is neither a rank algorithm of the scene nor current_score", who,
                          expr_rewrite_source(e, r), name);
                return PG_ERR_INVALID;
            }
            var_src->push_back(found);
        }
    return PG_OK;
}

int post_scratch(pg_ctx* ctx, const RecommendCall& c, uint32_t nq, PostScratch* ps) {
    const uint32_t n = nq * c.k;
    void* buf;
    int rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_local = al((size_t)n * 4), b_off = al((size_t)(nq + 1) * 4), b_err = al((size_t)kMaxQueries * 4);
    // variables of the widest expression + one f64 result per score rewrite
    int wide = std::max(c.nv, 1);
    const int n_rw = expr_num_rewrites(c.e);
    for (int r = 0; r < n_rw; ++r) wide = std::max(wide, expr_rewrite_num_vars(c.e, r));
    const size_t b_vars = al((size_t)(wide + n_rw) * n * 8);
    if ((rc = scratch_reserve(ctx, 8, b_local + b_off + b_err + b_vars, &buf))) return rc;
    ps->d_local = (uint32_t*)buf;
    ps->d_off = (uint32_t*)((char*)buf + b_local);
    ps->d_err = (uint32_t*)((char*)buf + b_local + b_off);
    ps->d_vars = (double*)((char*)buf + b_local + b_off + b_err);
    ps->c_rows = nullptr; ps->c_rel = nullptr; ps->c_emb = nullptr; ps->c_bail = nullptr;
    if (c.rerank.kind) {
        const size_t nc = (size_t)nq * c.rerank.candidates;
        const size_t b_rows = al(nc * 8), b_rel = al(nc * 8), b_emb = al(nc * c.t->dim * 4), b_bail = al((size_t)kMaxQueries * 4);
        if ((rc = scratch_reserve(ctx, 10, b_rows + b_rel + b_emb + b_bail, &buf))) return rc;
        ps->c_rows = (uint64_t*)buf;
        ps->c_rel = (double*)((char*)buf + b_rows);
        ps->c_emb = (float*)((char*)buf + b_rows + b_rel);
        ps->c_bail = (uint32_t*)((char*)buf + b_rows + b_rel + b_emb);
    }
    return PG_OK;
}

// The fusion itself: ScoreRewrite sources first, then RankScore, over n items whose planes lie rank_stride apart; d_vars holds
// (widest expression's variables + number of rewrites) x n doubles; d_err kMaxQueries flags (one per items_per_flag items)
int fuse_scores_enqueue_locked(pg_ctx* ctx, const pg_expr* e, const int* var_src, int nv, const float* d_recall, const float* d_rank,
                               size_t rank_stride, uint32_t n, uint32_t items_per_flag, double* d_vars, uint32_t* d_err, double* d_fused) {
    hipStream_t st = ctx->stream;
    int rc;
    VarSrc vs;
    // RankConfig.ScoreRewrite (rank_service.go:343-353): every source's expression over the scores as the algorithms left
    // them, all of them before any is written back; results stay f64 (AddAlgoScores) behind the variables
    const int n_rw = expr_num_rewrites(e);
    int wide = std::max(nv, 1);
    for (int r = 0; r < n_rw; ++r) wide = std::max(wide, expr_rewrite_num_vars(e, r));
    double* const d_rw = d_vars + (size_t)wide * n;
    const uint32_t bind_grid = (std::max(n, (uint32_t)kMaxQueries) + 255) / 256;
    bool zeroed = false;
    for (int r = 0, at = nv; r < n_rw; ++r) {
        const int nvr = expr_rewrite_num_vars(e, r);
        for (int i = 0; i < 32; ++i) vs.src[i] = i < nvr ? (int8_t)var_src[at + i] : (int8_t)-1;
        at += nvr;
        if (nvr > 0 || !zeroed) {
            bind_vars_kernel<<<bind_grid, 256, 0, st>>>(d_recall, d_rank, rank_stride, n, (uint32_t)nvr, vs, d_vars, d_err, nullptr, zeroed ? 0 : 1);
            PG_HIP(hipGetLastError());
            zeroed = true;
        }
        if ((rc = expr_rewrite_eval_enqueue_locked(ctx, e, r, d_vars, n, d_rw + (size_t)r * n, d_err, items_per_flag))) return rc;
    }
    for (int i = 0; i < 32; ++i) vs.src[i] = i < nv ? (int8_t)var_src[i] : (int8_t)-1;
    if (nv > 0) {
        bind_vars_kernel<<<bind_grid, 256, 0, st>>>(d_recall, d_rank, rank_stride, n, (uint32_t)nv, vs, d_vars, d_err, d_rw, zeroed ? 0 : 1);
        PG_HIP(hipGetLastError());
    } else if (!zeroed) {
        PG_HIP(hipMemsetAsync(d_err, 0, (size_t)kMaxQueries * 4, st));
    }
    return expr_eval_enqueue_locked(ctx, e, d_vars, n, d_fused, d_err, items_per_flag);
}

// RankScore fusion over the algorithms' score planes + ItemRankScore sort for requests [q0, q0 + nq) of the call
// (rank_service.go:339-363, sort/item_rank_score.go:26-32); ps.d_off must hold the uniform offsets
int post_fuse_sort_locked(pg_ctx* ctx, const RecommendCall& c, uint32_t q0, uint32_t nq, const PostScratch& ps) {
    const uint32_t n = nq * c.k;
    const size_t o = (size_t)q0 * c.k;
    hipStream_t st = ctx->stream;
    int rc;
    if ((rc = fuse_scores_enqueue_locked(ctx, c.e, c.var_src, c.nv, c.d_recall + o, c.d_rank + o, c.rank_stride, n, c.k, ps.d_vars, ps.d_err,
                                         c.d_fused + o)))
        return rc;
    if (c.pads) {
        mask_pads_kernel<<<(n + 255) / 256, 256, 0, st>>>(c.d_rows + o, n, c.d_rank + o, c.rank_stride, c.planes(), c.d_fused + o);
        PG_HIP(hipGetLastError());
    }
    return sort_dev_locked(ctx, c.d_fused + o, ps.d_off, nq, n, c.k, 1, c.d_order + o);
}

// DPPSort.doSort (sort/dpp_sort.go:271-351) on the sorted lists, in two halves with the embedding gather between them
// (local rows here, rows spread over shards in group.hip): the candidates = the first C entries (their global rows and
// relevance, normalised as dpp_norm_relevance_score asks) ...
int rerank_select_locked(pg_ctx* ctx, const RecommendCall& c, uint32_t q0, uint32_t nq, const PostScratch& ps) {
    const uint32_t C = c.rerank.candidates;
    const size_t o = (size_t)q0 * c.k;
    if (c.k < C) {
        set_error("recommend: the DPP stage wants %u candidates, the lists hold k = %u", C, c.k);
        return PG_ERR_UNSUPPORTED;
    }
    int rc;
    if ((rc = sorted_head_launch(ctx->stream, c.d_order + o, c.d_rows + o, c.d_fused + o, nq, c.k, C, ps.c_rows, ps.c_rel))) return rc;
    return dpp_norm_relevance_launch(ctx->stream, ps.c_rel, nq, C, c.rerank.dpp.norm_relevance_score, ps.c_bail);
}
// ... and KernelMatrix + DPPWithWindow for the whole batch at once over their embedding rows ps.c_emb [nq][C][dim]
int rerank_run_locked(pg_ctx* ctx, const RecommendCall& c, uint32_t q0, uint32_t nq, const PostScratch& ps) {
    const uint32_t C = c.rerank.candidates;
    int rc;
    if ((rc = dpp_run_locked(ctx, ps.c_emb, nullptr, ps.c_rel, nq, C, c.t->dim, 0, c.rerank.dpp.alpha, c.top_n, c.rerank.dpp.window,
                             c.rerank.dpp.normalize_emb, 1, 1, c.d_pick + (size_t)q0 * c.top_n, c.d_pick_cnt + q0)))
        return rc;
    if (c.rerank.dpp.norm_relevance_score)
        return dpp_bail_fix_launch(ctx->stream, ps.c_bail, nq, C, c.top_n, c.d_pick + (size_t)q0 * c.top_n, c.d_pick_cnt + q0);
    return PG_OK;
}

// the stages behind the recall for requests [q0, q0 + nq) of the call (caller holds ctx->mu; d_err is indexed from 0)
static int recommend_post_locked(pg_ctx* ctx, const RecommendCall& c, uint32_t q0, uint32_t nq, const PostScratch& ps) {
    const uint32_t n = nq * c.k;
    const size_t o = (size_t)q0 * c.k;
    int rc;
    if ((rc = rows_to_local_offsets_locked(ctx, c.t, c.d_rows + o, nq, c.k, ps.d_local, ps.d_off))) return rc;
    // RankAlgoList: every algorithm scores every candidate (rank_service.go:259-289 fans them out as goroutines)
    for (int a = 0; a < c.n_algos; ++a) {
        const RankAlgoRef& al = c.algos[a];
        float* out = c.d_rank + (size_t)c.plane0[a] * c.rank_stride + o;
        const float* users = c.d_queries + (size_t)q0 * c.t->dim;
        if ((rc = rank_algo_locked(ctx, al, c.t, users, c.d_ufids ? c.d_ufids + (size_t)q0 * c.ufid_stride : nullptr, ps.d_local, ps.d_off, nq, n,
                                   out, c.rank_stride)))
            return rc;
    }
    if ((rc = post_fuse_sort_locked(ctx, c, q0, nq, ps))) return rc;
    if (c.rerank.kind == 1) {
        if (c.t->rows < c.rerank.candidates) {
            set_error("recommend: the DPP stage wants %u candidates, the table has %llu rows", c.rerank.candidates, (unsigned long long)c.t->rows);
            return PG_ERR_UNSUPPORTED;
        }
        if ((rc = rerank_select_locked(ctx, c, q0, nq, ps))) return rc;
        if ((rc = gather_global_rows_launch(ctx->stream, c.t, ps.c_rows, nq * c.rerank.candidates, ps.c_emb))) return rc;
        if ((rc = rerank_run_locked(ctx, c, q0, nq, ps))) return rc;
    }
    return PG_OK;
}

int rank_algo_locked(pg_ctx* ctx, const RankAlgoRef& al, const pg_table* t, const float* d_user, const int32_t* d_ufids,
                     const uint32_t* d_cand, const uint32_t* d_off, uint32_t n_req, uint32_t n_items, float* d_out, size_t out_stride) {
    if (al.m->kind == PG_MODEL_DNN3) return rank_dnn3_dev_locked(ctx, al.m, t, d_user, d_cand, d_off, n_req, n_items, d_out, out_stride);
    if (al.irows) return rank_fm2t_irows_dev_locked(ctx, al.m, al.irows, d_user, d_ufids, d_cand, d_off, n_req, n_items, d_out);
    return rank_fm2t_rows_dev_locked(ctx, al.m, al.fs, al.item_field_cols, d_user, d_ufids, d_cand, d_off, n_req, n_items, d_out);
}

int recommend_enqueue(pg_ctx* ctx, const RecommendCall& c, PipeRun* r, bool first) {
    if (c.t->d_row_map) {                // the rank stage gathers the recalled rows from the same table: a view reports its source's ids
        set_error("recommend: the table is a filtered view (pg_table_view_create) — views serve the recall calls only");
        return PG_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    TimersScope quiet(ctx, c.timers);
    TableRead tr(c.t->rw);               // recall, rank and the DPP gather of one batch read one version of the table
    int rc;
    PostScratch ps;
    if ((rc = post_scratch(ctx, c, c.nq, &ps))) return rc;
    r->patched = false;
    // a swap / upload between this batch's first pass and its re-plan: start over on the new rows (one version per batch)
    if (!first && r->job.table_gen != c.t->generation.load(std::memory_order_relaxed)) first = true;
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
    if ((rc = recommend_post_locked(ctx, c, 0, c.nq, ps))) return rc;
    PG_HIP(hipMemcpyAsync(r->h_status + kExprFlagAt, ps.d_err, (size_t)c.nq * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipEventRecord(r->done, ctx->stream));
    return PG_OK;
}

int recommend_verify(pg_ctx* ctx, PipeRun* r, bool* ok, const RecommendCall* c) {
    std::lock_guard<std::mutex> g(ctx->mu);
    TableRead tr(r->job.t->rw);
    int rc;
    if ((rc = recall_job_check(&r->job, ok))) return rc;
    RecallJob& j = r->job;
    // (patching single requests in place is only sound on the version the rest of the batch came from; after a swap the
    // caller's re-enqueue restarts the whole batch)
    if (!*ok && !j.failed.empty() && j.failed.size() <= kMaxPatchQueries && j.nq > 1 &&
        j.table_gen == j.t->generation.load(std::memory_order_relaxed)) {
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
                if ((rc = recommend_post_locked(ctx, *c, q, 1, ps))) return rc;
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
    pg::ExprHold e_hold;
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
    PG_REQUIRE(m->n_out == 1, "pg_recommend_dnn3: a multi-output model needs its output names: serve it through a scene (pg_coalescer_create_scene)");
    pg_ticket* tk = new pg_ticket();
    tk->e_hold.take(e);
    int rc;
    if ((rc = pg::recommend_bind_vars(e, &rank_var, 1, &tk->var_src, "pg_recommend_dnn3"))) {
        delete tk;
        return rc;
    }
    pg::RecommendCall& c = tk->call;
    c.t = t; c.algos[0].m = m; c.n_algos = 1; c.e = e; c.var_src = tk->var_src.data(); c.nv = pg_expr_num_vars(e);
    c.d_queries = d_queries; c.nq = nq; c.k = k;
    c.d_rows = d_out_rows; c.d_recall = d_out_recall_scores; c.d_rank = d_out_rank_scores; c.rank_stride = (size_t)nq * k;
    c.d_fused = d_out_fused; c.d_order = d_out_order; c.d_count = d_out_count;
    c.pads = t->rows < k;
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

int pg_fuse_scores_dev(pg_ctx* ctx, const pg_expr* e, const char* const* plane_names, uint32_t n_planes, const float* d_rank,
                       size_t rank_stride, const float* d_recall, uint32_t n, double* d_fused) {
    PG_REQUIRE(ctx && e && (n_planes == 0 || (plane_names && d_rank)) && d_recall && d_fused, "pg_fuse_scores_dev: NULL argument");
    PG_REQUIRE(n_planes <= (uint32_t)pg::kMaxPlanes, "pg_fuse_scores_dev: %u planes (at most %d)", n_planes, pg::kMaxPlanes);
    if (n == 0) return PG_OK;
    std::vector<int> var_src;
    pg::ExprHold hold;
    hold.take(e);
    int rc;
    if ((rc = pg::recommend_bind_vars(e, plane_names, (int)n_planes, &var_src, "pg_fuse_scores_dev"))) return rc;
    const int nv = pg_expr_num_vars(e), n_rw = pg::expr_num_rewrites(e);
    int wide = std::max(nv, 1);
    for (int r = 0; r < n_rw; ++r) wide = std::max(wide, pg::expr_rewrite_num_vars(e, r));
    std::lock_guard<std::mutex> g(ctx->mu);
    void* buf;
    const size_t b_vars = (((size_t)(wide + n_rw) * n * 8) + 255) & ~(size_t)255;
    if ((rc = pg::scratch_reserve(ctx, 15, b_vars + (size_t)pg::kMaxQueries * 4, &buf))) return rc;
    double* const d_vars = (double*)buf;
    uint32_t* const d_err = (uint32_t*)((char*)buf + b_vars);
    if ((rc = pg::fuse_scores_enqueue_locked(ctx, e, var_src.data(), nv, d_recall, d_rank, rank_stride, n, 0u, d_vars, d_err, d_fused))) return rc;
    // (items_per_flag = 0: one flag for the call; read through the context's pinned status words)
    PG_HIP(hipMemcpyAsync(ctx->h_status + 321, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->h_status[321] != 0) {
        pg::set_expr_arith_error(e);
        return PG_ERR_ARITH;
    }
    return PG_OK;
}

int pg_recommend_end_timed(pg_ctx* ctx, pg_ticket* tk, uint32_t timeout_us, double* scan_ms) {
    PG_REQUIRE(ctx && tk, "pg_recommend_end_timed: NULL argument");
    if (timeout_us) {
        // poll the batch's event up to the deadline; the verification (and a possible re-plan) only starts once it has
        // completed, so a PG_ERR_TIMEOUT leaves the ticket exactly as it was
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(timeout_us);
        for (;;) {
            const hipError_t q = hipEventQuery(tk->run->done);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) {
                pg::set_error("pg_recommend_end_timed: %s", hipGetErrorString(q));
                return PG_ERR_DEVICE;
            }
            if (std::chrono::steady_clock::now() >= deadline) {
                pg::set_error("pg_recommend_end_timed: the batch did not complete within %u us (the ticket stays valid)", timeout_us);
                return PG_ERR_TIMEOUT;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    return pg_recommend_end(ctx, tk, scan_ms);
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
