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

// One entry of RankConf.RankAlgoList (service/rank/rank_service.go:259-289): a DNN3 over the table's rows, or an
// FM + two-tower whose item field ids are integer columns of `fs` keyed by the same rows.
constexpr int kMaxAlgos = 4;
struct RankAlgoRef {
    const pg_model* m = nullptr;
    const pg_features* fs = nullptr;
    int32_t item_field_cols[16] = {0};
    const pg_item_rows* irows = nullptr;   // FM + two-tower: the materialised item records (preferred over fs / columns)
};
// rank the candidates of n_req requests with one algorithm of the list (caller holds ctx->mu)
// (a multi-output DNN3 writes one plane per head: head o at d_out + o * out_stride; out_stride = 0: n_items)
int rank_algo_locked(pg_ctx* ctx, const RankAlgoRef& al, const pg_table* t, const float* d_user, const int32_t* d_ufids,
                     const uint32_t* d_cand, const uint32_t* d_off, uint32_t n_req, uint32_t n_items, float* d_out,
                     size_t out_stride = 0);
constexpr int kMaxPlanes = 12;         // score planes of a scene: one per single-output algorithm, n_out per multi-output one
// The diversity re-rank behind the sort (SortNames: [.., DPPSort], sort/dpp_sort.go:271-351): the first `candidates`
// entries of every sorted list are the DPP candidates, the page is DPPWithWindow's pick sequence among them.
struct RerankStage {
    int kind = 0;                      // 0 none, 1 DPPSort
    uint32_t candidates = 0;           // max(ctx.Size, CandidateCount) — fixed per call (callers keep top_n <= candidates)
    pg_dpp_options dpp{};              // alpha, window, normalize_emb, norm_relevance_score; topn is the call's top_n
};

// VectorRecall → rank with every algorithm of the list → RankScore fusion → ItemRankScore sort → (DPPSort) for nq
// requests of k candidates each; every pointer is a device pointer, layouts as pg_recommend_dnn3_dev.  var_src[i] = p:
// variable i of `e` is score plane p (an algorithm's name in RankAlgoList, or "<algo>_<output>" of a multi-output one,
// rank_service.go:315-319), -1: Item.Score (the recall score).
struct RecommendCall {
    bool timers = true;                // stage-timer events around the recall plan, its scan launches and the rank stage (TimersScope)
    const pg_table* t = nullptr;
    RankAlgoRef algos[kMaxAlgos];
    int n_algos = 0;
    const pg_expr* e = nullptr;
    const int* var_src = nullptr;
    int nv = 0;
    const float* d_queries = nullptr;
    const int32_t* d_ufids = nullptr;  // [nq][ufid_stride] user field ids of the FM + two-tower algorithms (each reads its first nuf)
    uint32_t ufid_stride = 0;
    uint32_t nq = 0, k = 0;
    uint64_t* d_rows = nullptr;
    float* d_recall = nullptr;
    float* d_rank = nullptr;           // planes() planes of rank_stride floats, each [nq][k]
    size_t rank_stride = 0;
    // score planes: algorithm a writes planes plane0[a] .. plane0[a] + (its model's outputs) - 1; n_planes = 0 means
    // one plane per algorithm in list order (no multi-output model in the list)
    int plane0[kMaxAlgos] = {0, 1, 2, 3};
    int n_planes = 0;
    int planes() const { return n_planes ? n_planes : n_algos; }
    double* d_fused = nullptr;
    uint32_t* d_order = nullptr;
    uint32_t* d_count = nullptr;       // optional
    bool pads = false;                 // the table has fewer than k rows: lists end in padding slots (masked before the sort)
    RerankStage rerank;
    uint32_t top_n = 0;                // re-rank: picks per request
    uint32_t* d_pick = nullptr;        // re-rank: [nq][top_n] positions in the sorted list
    uint32_t* d_pick_cnt = nullptr;    // re-rank: [nq]
};
// Resolve the expression's variables against the rank algorithms' names and "current_score" (module/item.go:189-212).
int recommend_bind_vars(const pg_expr* e, const char* const* names, int n_algos, std::vector<int>* var_src, const char* who);
// Enqueue the batch (first = true) or its next recall plan plus everything behind it (first = false, after a failed
// verification).  Takes ctx->mu for the duration of the enqueue only; never synchronises after the first use of a table.
int recommend_enqueue(pg_ctx* ctx, const RecommendCall& c, PipeRun* r, bool first);
// After r->done has completed: *ok = false → the recall plan did not hold (call recommend_enqueue(first = false) and wait
// again).  With *ok = true, r->h_status[kExprFlagAt + q] != 0 marks requests whose RankScore divided by zero.
// When the pilot threshold was too high for a few requests only, they are re-run here, synchronously and in place
// (c != NULL: recall and everything behind it; c == NULL: a recall-only job), r->patched is set and *ok = true.
int recommend_verify(pg_ctx* ctx, PipeRun* r, bool* ok, const RecommendCall* c = nullptr);
// The stages behind the rank as separate steps, for hosts that put something between them (group.hip: the score
// and embedding exchanges between shards).  All take the call's device buffers, requests [q0, q0 + nq), and the
// context's post-stage scratch (post_scratch; caller holds ctx->mu).
struct PostScratch {
    uint32_t *d_local, *d_off, *d_err;
    double* d_vars;
    // re-rank stage (scratch slot 10)
    uint64_t* c_rows;
    double* c_rel;
    float* c_emb;
    uint32_t* c_bail;
};
int post_scratch(pg_ctx* ctx, const RecommendCall& c, uint32_t nq, PostScratch* ps);
int post_fuse_sort_locked(pg_ctx* ctx, const RecommendCall& c, uint32_t q0, uint32_t nq, const PostScratch& ps);
int fuse_scores_enqueue_locked(pg_ctx* ctx, const pg_expr* e, const int* var_src, int nv, const float* d_recall, const float* d_rank,
                               size_t rank_stride, uint32_t n, uint32_t items_per_flag, double* d_vars, uint32_t* d_err, double* d_fused);
int rerank_select_locked(pg_ctx* ctx, const RecommendCall& c, uint32_t q0, uint32_t nq, const PostScratch& ps);
int rerank_run_locked(pg_ctx* ctx, const RecommendCall& c, uint32_t q0, uint32_t nq, const PostScratch& ps);

// misc.hip launchers (caller holds ctx->mu)
int rows_to_local_locked(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t n, uint32_t* d_local,
                         uint8_t* d_owned);
int uniform_offsets_locked(pg_ctx* ctx, uint32_t nq, uint32_t k, uint32_t* d_off);
int rows_to_local_offsets_locked(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t nq, uint32_t k, uint32_t* d_local,
                                 uint32_t* d_off);
// the first C entries of every sorted list → their global rows and relevance (fused score), [nq][C]
int sorted_head_launch(hipStream_t st, const uint32_t* d_order, const uint64_t* d_rows, const double* d_fused, uint32_t nq,
                       uint32_t k, uint32_t C, uint64_t* d_c_rows, double* d_c_rel);
// d_out[i][dim] = row d_global_rows[i] of `t` where the table holds it; other entries are left as they are
int gather_global_rows_launch(hipStream_t st, const pg_table* t, const uint64_t* d_global_rows, uint32_t n, float* d_out);
// The page of every request: entry order[q][pick[q][p]] (pick = NULL: p itself) of its list, p < top_n, as planes
// [nq][top_n]: rows u64 | fused f64 | recall f32 | n_algos x rank f32.  Entries beyond pick_cnt[q] are padding
// (row = UINT64_MAX, fused = NaN, recall = -inf, rank = 0).
size_t page_entry_bytes(int n_algos);
int page_launch(hipStream_t st, const uint32_t* d_order, const uint32_t* d_pick, const uint32_t* d_pick_cnt,
                const uint64_t* d_rows, const float* d_recall, const float* d_rank, size_t rank_stride, int n_algos,
                const double* d_fused, uint32_t nq, uint32_t k, uint32_t top_n, char* d_page);
// dpp.hip: dpp_norm_relevance_score (sort/dpp_sort.go:382-405) for nq lists of n relevance scores on the device, in
// the reference's operation order; mode 0 leaves them; d_bail[q] = 1 where the reference bails out ("all item score
// is zero": the items stay as they are)
int dpp_norm_relevance_launch(hipStream_t st, double* d_rel, uint32_t nq, uint32_t n, int mode, uint32_t* d_bail);
// picks of bailed requests := 0 .. top_n - 1 (the sorted list's own order)
int dpp_bail_fix_launch(hipStream_t st, const uint32_t* d_bail, uint32_t nq, uint32_t n, uint32_t top_n, uint32_t* d_pick,
                        uint32_t* d_pick_cnt);
// host form of the same normalisation (pg_dpp_ex and the coalescer's single-request DPP calls): returns false on bail-out
bool dpp_norm_relevance_host(const double* rel, uint32_t n, int mode, double* out);
// ssd.hip: SSDWithSlidingWindow for R requests of n candidates (d_cand [R][n] rows of t, d_rel [R][n] quality scores,
// d_out [R][T]); R > 1 needs ssd_batchable(d1, window).  Caller holds ctx->mu.
int ssd_run_locked(pg_ctx* ctx, const pg_table* t, const uint32_t* d_cand, const double* d_rel, uint32_t R, uint32_t n,
                   double gamma, uint32_t T, uint32_t window, int normalize_emb, int ensure_pos_similarity, int use_ssd_star,
                   uint32_t* d_out);
bool ssd_batchable(uint32_t d1, uint32_t window);
bool ssd_norm_quality_host(const double* rel, uint32_t n, int mode, double* out);

}  // namespace pg
