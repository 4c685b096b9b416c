// pipeline.hpp — a request batch that stays on the device from the user vectors to the sorted candidate lists,
// enqueued without a single host synchronisation and verified afterwards (pipeline.hip); shared with the request
// coalescer (coalescer.hip) and the shard group (group.hip).
#pragma once
#include "common.hpp"

namespace pg {

// Private resources of one in-flight batch: several may be queued on a context's stream at once, so whatever the
// host reads back later (status words, timing events) cannot live in the context.
struct RecommendCall;
struct PipeRun {
    RecallJob job;
    uint32_t* h_status = nullptr;      // pinned: [0 .. 256] recall status, [kExprFlagAt .. +256) RankScore flags per request
    std::vector<hipEvent_t> events;    // the recall's timing events
    hipEvent_t done = nullptr;         // recorded behind the batch's last command
    bool patched = false;              // the last verification re-ran a few requests in place: device outputs changed after `done`
};
constexpr uint32_t kExprFlagAt = 320;
constexpr uint32_t kPipeStatusWords = 640;

int pipe_run_acquire(pg_ctx* ctx, PipeRun** out);      // from the context's pool (creates on demand)
void pipe_run_release(pg_ctx* ctx, PipeRun* r);

// VectorRecall → DNN3 rank → RankScore fusion → ItemRankScore sort for nq requests of k candidates each; every
// pointer is a device pointer, layouts as pg_recommend_dnn3_dev.  var_src[i] = 1: variable i of `e` is the model's
// score, 0: Item.Score (the recall score).
struct RecommendCall {
    const pg_table* t = nullptr;
    const pg_model* m = nullptr;
    const pg_expr* e = nullptr;
    const int* var_src = nullptr;
    int nv = 0;
    const float* d_queries = nullptr;
    uint32_t nq = 0, k = 0;
    uint64_t* d_rows = nullptr;
    float* d_recall = nullptr;
    float* d_rank = nullptr;
    double* d_fused = nullptr;
    uint32_t* d_order = nullptr;
    uint32_t* d_count = nullptr;       // optional
};
// Resolve the expression's variables against the rank algorithm's name and "current_score" (module/item.go:189-212).
int recommend_bind_vars(const pg_expr* e, const char* rank_var, std::vector<int>* var_src, const char* who);
// Enqueue the batch (first = true) or its next recall plan plus everything behind it (first = false, after a failed
// verification).  Takes ctx->mu for the duration of the enqueue only; never synchronises after the first use of a table.
int recommend_enqueue(pg_ctx* ctx, const RecommendCall& c, PipeRun* r, bool first);
// After r->done has completed: *ok = false → the recall plan did not hold (call recommend_enqueue(first = false) and wait
// again).  With *ok = true, r->h_status[kExprFlagAt + q] != 0 marks requests whose RankScore divided by zero.
// When the pilot threshold was too high for a few requests only, they are re-run here, synchronously and in place
// (c != NULL: recall and everything behind it; c == NULL: a recall-only job), r->patched is set and *ok = true.
int recommend_verify(pg_ctx* ctx, PipeRun* r, bool* ok, const RecommendCall* c = nullptr);

// misc.hip launchers (caller holds ctx->mu)
int rows_to_local_locked(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t n, uint32_t* d_local,
                         uint8_t* d_owned);
int uniform_offsets_locked(pg_ctx* ctx, uint32_t nq, uint32_t k, uint32_t* d_off);

}  // namespace pg
