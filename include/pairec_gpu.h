/*
 * pairec_gpu.h — C ABI of libpairec_gpu.so, the MI355X (gfx950) engine behind pairec's
 * rank + recall hot path.
 *
 * This is the drop-in boundary: every entry point replaces one network hop of the reference
 * (alibaba/pairec, Go).  A cgo shim implements algorithm.IAlgorithm,
 * recall.Recall and sort.ISort on top of these calls.  Signatures use plain pointers and sizes;
 * no C++ or torch types cross the boundary.
 *
 * Conventions
 *   - every function returns 0 on success, a negative pg_status on failure; pg_last_error()
 *     returns a thread-local message.  Nothing throws or aborts across the ABI (the reference's
 *     rank goroutines have no recover(), service/rank/rank_service.go:265-288).
 *   - all entry points are re-entrant: the reference calls IAlgorithm.Run concurrently from one
 *     goroutine per batch × per algo (rank_service.go:264-289).  Calls on one pg_ctx are serialised
 *     on that context's HIP stream.
 *   - "_dev" variants take device pointers (HBM-resident inputs/outputs, used to chain stages and
 *     by bench.py); the plain variants take host pointers (what the cgo shim passes) and copy.
 *   - row ids are uint32 table row indices (+ a uint64 per-table row_offset for sharded tables);
 *     item-id strings stay on the host side (module.ItemId is a string, module/item.go:13).
 */
#ifndef PAIREC_GPU_H
#define PAIREC_GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pg_ctx pg_ctx;
typedef struct pg_table pg_table;      /* HBM-resident embedding table  (module.VectorDao backend) */
typedef struct pg_model pg_model;      /* rank model weights            (algorithm/eas model)      */
typedef struct pg_expr pg_expr;        /* compiled RankScore expression (utils/ast)                */

typedef enum {
    PG_OK = 0,
    PG_ERR_INVALID = -1,     /* bad argument */
    PG_ERR_DEVICE = -2,      /* HIP runtime error (message carries hipGetErrorString) */
    PG_ERR_NOMEM = -3,
    PG_ERR_UNSUPPORTED = -4, /* shape outside what the kernels are built for */
    PG_ERR_ARITH = -5,       /* expression: division by zero / modulo by zero (reference panics) */
    PG_ERR_PARSE = -6,       /* expression: lexer "symbol error" (utils/ast/parse.go:125-133) */
    PG_ERR_EMPTY = -8,       /* pg_table_view_create: no row passes the filter (a result, not a misconfiguration) */
    PG_ERR_TIMEOUT = -7      /* the call's deadline passed (algorithm/eas/client.go:53-58: 100 ms default per predict); the
                                work it belonged to still completes for the other callers of its batch */
} pg_status;

/* Precision of a rank model's two matrix layers.  The reference hands model outputs on as fp32 widened to f64
 * (algorithm/eas/easyrec_response.go:479-483, eas/tf_response.go:55-59).
 *   PG_PREC_F32    fp32 MFMA, bit-defined (a k-ordered fmaf chain): the specification.
 *   PG_PREC_BF16   operands rounded to bf16, fp32 accumulation: fastest; scores within ~4e-5 of the fp32 path.
 *   PG_PREC_BF16X3 "split bf16": every operand as hi + lo bf16, three products per term into the fp32 accumulator, nothing
 *                  else rounded: scores within 1e-5 of PG_PREC_F32 (observed ~1e-6) at the bf16 matrix pipe's speed / 3. */
typedef enum { PG_PREC_F32 = 0, PG_PREC_BF16 = 1, PG_PREC_BF16X3 = 2 } pg_prec;
typedef enum { PG_MODEL_DNN3 = 1, PG_MODEL_FM_TWOTOWER = 2, PG_MODEL_DNN3_MULTI = 3 } pg_model_kind;

const char* pg_last_error(void);
const char* pg_version(void);

/* ---- context --------------------------------------------------------------------------------
 * One context = one GPU + one HIP stream.  `stream` may be an existing hipStream_t (e.g. a dedicated stream
 * torch.distributed's collectives are ordered on) or NULL to create a private one.  The null stream (handle 0,
 * torch's default stream) cannot be adopted — NULL always means "private": a host that mixes its own device work
 * with library calls creates a stream of its own and passes it here (pairec_amd/dist.py shard_context). */
/* gfx950 devices this process can see (what a host sizes pg_group_create / its replica list with; the reference has no
 * counterpart — its FAISS / EAS endpoints are URLs in recconf, algorithm/eas/model.go:38-60). */
int pg_device_count(int* out);
int pg_init(int device, void* stream, pg_ctx** out);
int pg_shutdown(pg_ctx* ctx);
int pg_synchronize(pg_ctx* ctx);
/* Developer / test knobs of one context (defaults come from PG_* environment variables read once, in pg_init):
 * "no_pilot", "recall_exact", "screen_min", "pilot_fraction", "chunk_growth", "seed_rows", "pilot_growth",
 * "pilot_sigmas", "debug_scan", "rank_no_ws", "sort_lds", and for the 4-bit screen of batches of <= 4 queries
 * (csrc/recall_i4.hip) "no_screen_i4", "i4_min_rows" (default 2^22), "i4_max_lambda", and for the threshold refinement inside the pilot plan's
 * full pass "no_refine", "refine_min_rows" (default 2^24), and for the threshold model that replaces the pilot sample
 * once a table has seen >= 1024 queries of one K (plan 0) "no_predict", "predict_sigmas" (default 4.5),
 * "predict_min_rows" (default 2^22), and for the 256-query screen "screen_early_share" (default 604: the share
 * x 1024 of a SIMD's blocks its older wave takes; 512 = even), "screen_early_share_narrow" (the <= 128-query kernels, default
 * 512); for the squared-Euclidean recall "l2_exact", "l2_max_slack" (default 1.0); for pg_recall_topk_where the compact route's
 * limits "where_compact_max_rows" (default 2^23; half of it for batches of <= 4 queries) and "where_compact_min_ratio"
 * (default 8: at most an eighth of the table); round 6: for the 1 … 64-query passes on the 4-bit shadow through the matrix pipe
 * (csrc/recall_i4m.hip) "no_screen_i4m", "i4m_min_queries" (default 1), "i4m_max_queries" (64), "i4m_max_lambda", "i4m_max_pairs";
 * for crowded tables "max_rec_scale", "no_r2", "r2_min_factor", "predict_max_factor"; for the coalescer "coalescer_rejoin",
 * "coalescer_rejoin_us_per_caller"; and which sort a call with few lists takes (csrc/split_sort.hpp): "rank_sort_max" (default 32
 * lists) with "rank_sort_work" (lists x items^2 <= 7e7: counting ranks), "split_sort_max" (default 96 lists of 1025 … 8192 items:
 * runs sorted wave by wave over the chip; 0 = never); "stage_timers" (default 1; 0: this context's direct calls record no HIP events
 * around the recall plan, its scan launches and the rank stage — pg_stats' last_*_ms and pg_last_scan_kernel_ms stop moving, a
 * small batch's step gets 40-60 us shorter; a coalescer's batches never record them, see there).  value is parsed as a number. */
int pg_set_option(pg_ctx* ctx, const char* name, const char* value);
int pg_device_malloc(pg_ctx* ctx, size_t bytes, void** out);
int pg_device_free(pg_ctx* ctx, void* p);
int pg_memcpy_h2d(pg_ctx* ctx, void* dst, const void* src, size_t bytes);
int pg_memcpy_d2h(pg_ctx* ctx, void* dst, const void* src, size_t bytes);

/* ---- embedding tables -----------------------------------------------------------------------
 * Replaces module.VectorDao.VectorString (module/vector_dao.go:13-15) and its six remote
 * back-ends: rows live in HBM as row-major fp32 [rows][dim].  dim must be a multiple of 64. */
int pg_table_create(pg_ctx* ctx, uint64_t rows, uint32_t dim, uint64_t row_offset, pg_table** out);
int pg_table_destroy(pg_ctx* ctx, pg_table* t);
/* deterministic synthetic fill: global row = row_offset + local row */
int pg_table_fill_synthetic(pg_ctx* ctx, pg_table* t, uint64_t seed, int normalize);
/* i.i.d. N(0, sigma^2) elements, unnormalised rows — the second benchmark distribution (trained embeddings look
 * Gaussian; the uniform rows of SURVEY.md 8d are the int8 screen's best case).  Defined by the device's own fp64
 * log / cos: reproducible run to run, compared in tests against downloaded rows. */
int pg_table_fill_gaussian(pg_ctx* ctx, pg_table* t, uint64_t seed, float sigma);
/* clustered rows — the third benchmark distribution (trained embeddings cluster: the tables behind
 * service/recall/hologres_vector_recall.go:23): n_centres centres on the unit sphere, a row = its centre + noise of norm
 * ~ sigma, normalised.  No transcendental in it: oracle/oracle.c regenerates any slice bit for bit. */
int pg_table_fill_mixture(pg_ctx* ctx, pg_table* t, uint64_t seed, uint32_t n_centres, float sigma);
int pg_table_upload(pg_ctx* ctx, pg_table* t, uint64_t row0, uint64_t nrows, const float* host_rows);
int pg_table_download(pg_ctx* ctx, const pg_table* t, uint64_t row0, uint64_t nrows, float* host_rows);
/* atomically exchange the contents of two tables of equal shape (the analogue of the Hologres
 * partition hot-swap, module/vector_hologres_dao.go:40-61) */
int pg_table_swap(pg_ctx* ctx, pg_table* a, pg_table* b);
int pg_table_info(const pg_table* t, uint64_t* rows, uint32_t* dim, uint64_t* row_offset);
/* embedding lookup by row (the VectorString analogue): out[n][dim] fp32 */
int pg_table_gather(pg_ctx* ctx, const pg_table* t, const uint32_t* rows, uint32_t n, float* out);

/* ---- recall: exact inner-product top-K ------------------------------------------------------
 * Replaces FaissModel.Run → VectorClient.Search (algorithm/faiss/model.go:29-31,
 * vector_client.go:32-41; VectorRequest{k, vector} → VectorReply{retval, scores},
 * vectorretrieval.proto:11-20) and the Hologres pm_approx_inner_product_distance ORDER BY desc
 * LIMIT n query (service/recall/hologres_vector_recall.go:23).
 *   score(row) = chain_k fmaf(x[row][k], q[k], acc)  (k ascending, fp32) — independent of nq.
 *   order: score descending (IEEE totalOrder, NaN last), then row ascending.
 * queries: [nq][dim] fp32, nq <= 256 per call (<= 32 when dim > 128); one call = one table pass.
 * Finite tables of dim 128 / 64 are scanned with an int8 / bf16 MFMA screen over a quantised shadow of the rows
 * (a rigorous bound of every score) followed by exact re-scoring of the survivors, everything else with the
 * exact fp32-MFMA scan (groups of 64 queries per launch) — results are identical bit for bit.  out_rows: [nq][k] global row ids (row_offset + local), out_scores:
 * [nq][k].  If the table has fewer than k rows the tail is filled with
 * row = UINT64_MAX, score = -inf and *out_count (optional) receives the valid count. */
int pg_recall_topk(pg_ctx* ctx, const pg_table* t, const float* queries, uint32_t nq, uint32_t k,
                   uint64_t* out_rows, float* out_scores, uint32_t* out_count);
int pg_recall_topk_dev(pg_ctx* ctx, const pg_table* t, const float* d_queries, uint32_t nq,
                       uint32_t k, uint64_t* d_out_rows, float* d_out_scores, uint32_t* out_count);
/* HologresVectorRecallV2 (service/recall/hologres_vector_recall_v2.go:23,96-206): the k rows of SMALLEST squared Euclidean
 * distance to each query, ascending, the distance as the item's score (:181-189).  Exact (the reference's Proxima index is
 * approximate): d = fmaf(-2, ip, |x|^2 + |q|^2), each sum a k-ascending fp32 fmaf chain; ties by row ascending; slots beyond
 * the table's rows carry row UINT64_MAX and distance +inf.  dim 64 or 128.  A dim-128 table with an int8 shadow is served from
 * it — one integer cutoff per 32-row block where the rows have (nearly) one norm, a per-row test otherwise (knob "l2_max_slack");
 * suspects are re-scored exactly — any other table (and knob "l2_exact") by the exact scan over the fp32 rows. */
int pg_recall_topk_l2(pg_ctx* ctx, const pg_table* t, const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows,
                      float* out_dist, uint32_t* out_count);
int pg_recall_topk_l2_dev(pg_ctx* ctx, const pg_table* t, const float* d_queries, uint32_t nq, uint32_t k, uint64_t* d_out_rows,
                          float* d_out_dist, uint32_t* out_count);
/* I2IVectorRecall.GetCandidateItems (service/recall/item_2_item_vector_racall.go:51-152): the embeddings of the
 * trigger items (context parameter "item_id" → dao.VectorString) are the queries — rows `trigger_rows[n]` of
 * `trigger_table` (same dim as `t`; usually the same table) against `t`.  Outputs as pg_recall_topk; the trigger
 * item itself is not excluded (neither does the reference's SQL). */
int pg_i2i_recall(pg_ctx* ctx, const pg_table* trigger_table, const uint32_t* trigger_rows, uint32_t n,
                  const pg_table* t, uint32_t k, uint64_t* out_rows, float* out_scores, uint32_t* out_count);
/* OnlineVectorRecall.GetCandidateItems (service/recall/online_vector_recall.go:73-155): user features → the vector
 * model's user embedding → its FaissNeighNum = k nearest items (TorchrecEmbeddingItemsResponse: match_item_scores +
 * item_ids, algorithm/eas/easyrec_response.go:700-734).  `m` is a PG_MODEL_FM_TWOTOWER whose user tower produces the
 * embedding (pg_fm2t_user_embedding), `item_emb` the item-tower outputs as a table of dim t_out. */
int pg_online_vector_recall(pg_ctx* ctx, const pg_model* m, const pg_table* item_emb, const float* user_vecs,
                            uint32_t n_req, uint32_t k, uint64_t* out_rows, float* out_scores, uint32_t* out_count);
/* merge `nlists` sorted/unsorted (row,score) lists of `per_list` entries per query into the global
 * top-k (multi-GPU: the lists are the all-gathered per-shard results).  Layout [nq][nlists][per_list]. */
int pg_topk_merge_dev(pg_ctx* ctx, const uint64_t* d_rows, const float* d_scores, uint32_t nq,
                      uint32_t nlists, uint32_t per_list, uint32_t k, uint64_t* d_out_rows,
                      float* d_out_scores);

/* global row ids (as returned by recall/merge) → local row indices of table `t` for the rank stage;
 * d_owned (optional, uint8 per entry) receives 1 where the row lives in this shard, else 0 and
 * the local index is written as 0.  UINT64_MAX padding is "not owned". */
int pg_rows_to_local_dev(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t n,
                         uint32_t* d_local, uint8_t* d_owned);

/* ---- rank: model predict --------------------------------------------------------------------
 * Replaces EasModel.Run / TFservingModel.Run (algorithm/eas/model.go:197-222,
 * algorithm/tfserving/model.go:30-55): the DNN / FM forward that the reference ships to a
 * remote PAI-EAS / TF-Serving process, one call per batch of BatchCount=100 items
 * (service/rank/rank_service.go:163-166).  Here one call scores any number of requests.
 *
 * PG_MODEL_DNN3 blob (little-endian fp32 unless noted):
 *   u32 d_user, d_item, h1, h2;  w1[(d_user+d_item)][h1]; b1[h1]; w2[h1][h2]; b2[h2]; w3[h2]; b3
 *   score = sigmoid( w3 · relu( W2ᵀ relu( W1ᵀ [user ‖ item_row] + b1 ) + b2 ) + b3 )
 *   shapes: d_user 1..4096; d_item 64 or 128 (= the table's dim); (h1, h2) in {128-128, 256-128, 256-256, 512-256,
 *   1024-512}; the benchmark shape [128+128]-512-256 in bf16 runs on the weights-stationary kernel
 * PG_MODEL_DNN3_MULTI blob — a multi-output model: n_out heads on ONE shared trunk, the shape of the reference's own fixtures
 * (EasyrecResponse.multiValModule, algorithm/eas/easyrec_response.go:35-70: probs_ctr / probs_cvr of one PAI-EAS model;
 * RankService writes them as "<algo>_<output>", service/rank/rank_service.go:315-319):
 *   u32 d_user, d_item, h1, h2, n_out (1..8);  w1; b1; w2; b2 as above; w3[h2][n_out]; b3[n_out]
 *   score_o = sigmoid( w3[:, o] · h2 + b3[o] ) — every head has exactly the arithmetic of a PG_MODEL_DNN3 with that column
 *   (in PG_PREC_F32, bit for bit); ONE gather and ONE trunk per item whatever n_out.  pg_model_load stores it as a
 *   PG_MODEL_DNN3 with pg_model_num_outputs() = n_out: pg_rank_dnn3[_dev] then write n_out planes, out_scores[o * n_items + i]
 *   (n_items = req_offsets[n_req], resp. the n_items argument), and pg_coalescer_rank_dnn3 n_out planes of its n candidates.
 * Exported weights: every matrix is plain row-major [in][out] fp32 exactly as a Dense layer's kernel is saved;
 * tools/pack_model.py builds either blob from an .npz of such arrays.
 * PG_MODEL_FM_TWOTOWER blob:
 *   u32 n_user_fields, n_item_fields, k, d_user, t_h1, t_out, vocab; f32 fm_b;
 *   uw1[d_user][t_h1]; ub1; uw2[t_h1][t_out]; ub2; iw1[nif*k][t_h1]; ib1; iw2[t_h1][t_out]; ib2;
 *   then per field f (user fields first): emb_f[vocab][k], lin_f[vocab]
 *   score = sigmoid( y_fm + <user_tower(user), item_tower(concat item field embeddings)> )
 *   shapes: 1..16 user fields; n_item_fields x k = 128 with k in {8, 16, 32}; towers (t_h1, t_out, k) in
 *   {256-64 k16, 256-64 k32, 128-64 k8, 512-128 k16}
 * Summation orders are specified in DESIGN.md §5 (they are what makes PG_PREC_F32 bit-reproducible).
 */
int pg_model_load(pg_ctx* ctx, pg_model_kind kind, pg_prec prec, const void* blob, size_t len,
                  pg_model** out);
int pg_model_destroy(pg_ctx* ctx, pg_model* m);
/* outputs per item: 1, or n_out of a PG_MODEL_DNN3_MULTI */
int pg_model_num_outputs(const pg_model* m, uint32_t* out);

/* DNN3: R requests; request r has user vector user_vecs[r][d_user] and candidates
 * cand_rows[req_offsets[r] .. req_offsets[r+1]) (local row indices into `t`).  out_scores is fp32
 * per candidate, request order preserved (response.AlgoResponse order contract,
 * rank_service.go:312-335). */
int pg_rank_dnn3(pg_ctx* ctx, const pg_model* m, const pg_table* t, const float* user_vecs,
                 const uint32_t* cand_rows, const uint32_t* req_offsets, uint32_t n_req,
                 float* out_scores);
int pg_rank_dnn3_dev(pg_ctx* ctx, const pg_model* m, const pg_table* t, const float* d_user_vecs,
                     const uint32_t* d_cand_rows, const uint32_t* d_req_offsets, uint32_t n_req,
                     uint32_t n_items, float* d_out_scores);
/* FM + two-tower: user_field_ids [n_req][n_user_fields], item_field_ids [n_items][n_item_fields] */
int pg_rank_fm2t(pg_ctx* ctx, const pg_model* m, const float* user_vecs,
                 const int32_t* user_field_ids, const int32_t* item_field_ids,
                 const uint32_t* req_offsets, uint32_t n_req, float* out_scores);
int pg_rank_fm2t_dev(pg_ctx* ctx, const pg_model* m, const float* d_user_vecs,
                     const int32_t* d_user_field_ids, const int32_t* d_item_field_ids,
                     const uint32_t* d_req_offsets, uint32_t n_req, uint32_t n_items,
                     float* d_out_scores);

/* user-tower output of the two-tower model: out [n_req][t_out] = uw2' relu(uw1' P(u) + ub1) + ub2 —
 * the user embedding of online_vector_recall.go:97-109 / embedding_service.go:127 */
int pg_fm2t_user_embedding(pg_ctx* ctx, const pg_model* m, const float* user_vecs, uint32_t n_req, float* out);
int pg_fm2t_user_embedding_dev(pg_ctx* ctx, const pg_model* m, const float* d_user_vecs, uint32_t n_req, float* d_out);

/* ---- rank: score fusion (RankConfig.RankScore) ----------------------------------------------
 * Replaces ast.GetExpAST + ExprASTResult (utils/ast/ast.go:215-268,368-389): compile once,
 * evaluate per item in fp64 on the device.  Variables are bound by position: pg_expr_var_name(i)
 * names column i of `vars` ([n_vars][n_items] fp64, column-major per variable). */
int pg_expr_compile(const char* source, pg_expr** out);
int pg_expr_free(pg_expr* e);
int pg_expr_num_vars(const pg_expr* e);
/* GetExpASTWithType (utils/ast/ast.go:338-343): ast_type NULL, "" or anything but "antlr" = pg_expr_compile.  "antlr" selects the
 * reference's second evaluator (go-antlr-valuate v0.0.4, un-vendored); the engine serves the SUBSET its tests pin
 * (utils/ast/ast_test.go:30-56,90-167,213-300): + - * / ^ (^ = math.Pow, above * /, above + -), parentheses, unary minus,
 * numbers, ${name}, maxIndex(${list}) / maxValue(${list}) (antlr_functions.go:34-66) — `/` is float division (no panic).
 * Everything else returns PG_ERR_UNSUPPORTED naming the construct.  A list function becomes a variable called
 * "maxIndex(name)" / "maxValue(name)" (pg_expr_var_name) that the caller fills per item from the list property.
 * ExprASTResultByAntlr's error rule — an item that lacks ANY variable of the expression scores 0 (ast.go:374-383) — is the
 * caller's to apply (pg_expr_is_antlr tells it to). */
int pg_expr_compile_typed(const char* source, const char* ast_type, pg_expr** out);
int pg_expr_is_antlr(const pg_expr* e);
/* RankConfig.ScoreRewrite (recconf/recconf.go:743; service/rank/rank_service.go:296-306,343-353): a map source → expression.
 * Per item the reference evaluates EVERY source's expression over the item as the algorithms left it, collects the results
 * in a map, writes them back with Item.AddAlgoScores (overwriting / adding algorithm scores named `source`) and only then
 * evaluates RankScore.  Attach the scene's rewrites to its compiled RankScore; every pipeline that fuses scores with that
 * expression (pg_recommend_*, the scene / group coalescers, pg_group_*) then evaluates them first, on the device, in f64:
 * a RankScore variable naming a source reads the rewritten score, a rewrite's own variables read the un-rewritten ones
 * (the scene's "<algo>" / "<algo>_<output>" planes and current_score).  exprs[i] == NULL: the source's expression did not
 * compile — the reference logs it and scores 0 (rank_service.go:299-303,349-351).  The expressions are copied.
 * pg_expr_eval[_dev] evaluate the bare expression over caller-made variables and know nothing of rewrites.  n = 0 removes.
 * ATTACH BEFORE CREATING ANY PIPELINE FROM THE EXPRESSION: a coalescer (for its lifetime) and a batch begun and not yet ended
 * hold variable bindings sized for the rewrites present when they were made — while there is such a holder the call fails
 * with PG_ERR_INVALID and changes nothing. */
int pg_expr_set_score_rewrites(pg_expr* rank_score, uint32_t n, const char* const* sources, const pg_expr* const* exprs);
const char* pg_expr_var_name(const pg_expr* e, int i);
int pg_expr_eval(pg_ctx* ctx, const pg_expr* e, const double* vars, uint32_t n_items,
                 double* out_scores);
int pg_expr_eval_dev(pg_ctx* ctx, const pg_expr* e, const double* d_vars, uint32_t n_items,
                     double* d_out_scores);

/* float32 model outputs → float64 AlgoResponse scores, the widening every response decoder of
 * the reference performs (algorithm/eas/easyrec_response.go:479-483, eas/tf_response.go:55-59,
 * tfserving/response.go:51-64; recall: vector_recall.go:98). */
int pg_widen_f32_dev(pg_ctx* ctx, const float* d_in, uint32_t n, double* d_out);

/* The fusion step of RankService.Rank (service/rank/rank_service.go:339-363) alone, for a host that runs the stages itself
 * (pairec_amd/dist.py's sharded step between its collectives): n items, n_planes model-score planes d_rank[p * rank_stride + i]
 * (float32, widened as the decoders do) named plane_names[p] ("<algo>" or "<algo>_<output>"), d_recall[i] = Item.Score
 * ("current_score").  RankConfig.ScoreRewrite attached to `rank_score` (pg_expr_set_score_rewrites) is evaluated first, from
 * the un-rewritten planes (rank_service.go:343-353).  d_fused[i] = the RankScore, f64.  A division by zero anywhere is
 * PG_ERR_ARITH (the call synchronises to learn it); a variable that names neither a plane, a rewrite source nor
 * current_score PG_ERR_INVALID. */
int pg_fuse_scores_dev(pg_ctx* ctx, const pg_expr* rank_score, const char* const* plane_names, uint32_t n_planes,
                       const float* d_rank, size_t rank_stride, const float* d_recall, uint32_t n, double* d_fused);

/* ---- sort -----------------------------------------------------------------------------------
 * Replaces ItemRankScoreSort (descending, sort/item_rank_score.go:26-32) and ItemScoreSort
 * (ascending, sort/item_score.go:36-41): out_order[i] = index of the i-th item.  Segmented:
 * seg_offsets[n_seg+1] delimits independent requests.  Ties keep input order; NaN last. */
int pg_sort_scores(pg_ctx* ctx, const double* scores, const uint32_t* seg_offsets, uint32_t n_seg,
                   int descending, uint32_t* out_order);
/* max_segment: upper bound on a segment's length (sizes the scratch of the > 8192-item path);
 * 0 = unknown (n_items is assumed). */
int pg_sort_scores_dev(pg_ctx* ctx, const double* d_scores, const uint32_t* d_seg_offsets,
                       uint32_t n_seg, uint32_t n_items, uint32_t max_segment, int descending,
                       uint32_t* d_out_order);

/* ---- DPP diversity re-rank ------------------------------------------------------------------
 * Replaces DPPSort.KernelMatrix + DPPWithWindow (sort/dpp_sort.go:372-551).  Candidates are rows
 * of `t` (embeddings are L2-normalised in fp64 when normalize_emb != 0), rel = relevance scores
 * (Item.Score).  out_idx receives min(topn, n) indices into the candidate list. */
int pg_dpp(pg_ctx* ctx, const pg_table* t, const uint32_t* cand_rows, const double* rel, uint32_t n,
           double alpha, uint32_t topn, uint32_t window, int normalize_emb, uint32_t* out_idx,
           uint32_t* out_count);

/* The same with every switch of DPPSort.KernelMatrix (sort/dpp_sort.go:372-475):
 *   norm_relevance_score  the dpp_norm_relevance_score experiment parameter (:382-405): 0 none, 1 z-score
 *                         (stat.PopMeanVariance / StdScore), 2 min-max into [1e-6, 1] with max = first, min = last
 *                         candidate.  When the reference bails out ("all item score is zero") the call returns
 *                         PG_ERR_ARITH and the caller keeps the items unchanged, as DPPSort.doSort does.
 *   has_table = 1         embeddings are rows of `t` (DPPSortConfig.TableName set), L2-normalised when normalize_emb;
 *                         hook_emb (optional, [n][hook_dim] fp64: what the functions registered with
 *                         RegisterEmbeddingHook return, :52-58,362) is prepended and the row re-normalised (:419-421);
 *                         always followed by "append 1, scale 1/sqrt 2" (:428-430).
 *   has_table = 0         hook embeddings only (:434-447): normalised when normalize_emb; ensure_pos_similarity
 *                         (DPPSortConfig.EnsurePositiveSim) appends 1 and scales by 1/sqrt 2, otherwise appends 0.
 * out_relevance (optional, [n]) receives the relevance scores as used ("dpp_relevance_score", :410). */
typedef struct {
    double   alpha;
    uint32_t topn, window;
    int      normalize_emb, ensure_pos_similarity, norm_relevance_score, has_table;
    uint32_t hook_dim;
} pg_dpp_options;
int pg_dpp_ex(pg_ctx* ctx, const pg_table* t, const uint32_t* cand_rows, const double* rel, uint32_t n,
              const pg_dpp_options* opt, const double* hook_emb, uint32_t* out_idx, uint32_t* out_count,
              double* out_relevance);

/* ---- SSD diversity re-rank -------------------------------------------------------------------
 * Replaces SSDSort.SSDWithSlidingWindow (sort/ssd_sort.go:346-486) and the embedding treatment of
 * loadEmbeddingCache (:246-252).  Candidates are rows of `t`, given in score-descending order as
 * SSDSort.doSort leaves them (:296); rel = Item.Score.  normalize_emb / ensure_pos_similarity /
 * use_ssd_star / window / gamma are the SSDSortConfig fields of the same names (recconf.go:980-1000);
 * norm_quality_score is the ssd_norm_quality_score experiment parameter (0 none, 1 z-score, 2 min-max).
 * out_idx (capacity n) receives min(topn, n) indices into the candidate list; when the reference would
 * return the items unchanged ("all item score are zeros") it receives 0..n-1 and *out_count = n.
 * out_quality (optional, [n]) receives the normalised quality scores ("ssd_quality_score"). */
int pg_ssd(pg_ctx* ctx, const pg_table* t, const uint32_t* cand_rows, const double* rel, uint32_t n,
           double gamma, uint32_t topn, uint32_t window, int normalize_emb, int ensure_pos_similarity,
           int norm_quality_score, int use_ssd_star, uint32_t* out_idx, uint32_t* out_count,
           double* out_quality);

/* ---- item features: typed columns in HBM, assembled by row ----------------------------------
 * Replaces the per-request host boxing of EasyrecAlgoDataGenerator.AddFeatures / GeneratorAlgoData
 * (service/rank/algo_data.go:223-306): one column per feature name (the "context features" of
 * easyrec_predict.proto:150-212), an item that lacks a feature reads the column default
 * (feature.defaultValue, algo_data.go:154-171 — the Go zero value of the column's type).  Columns are
 * keyed by item row; a row index >= rows (UINT32_MAX by convention) means "item without the feature".
 * String features are dictionary-encoded to integer ids by the caller. */
typedef struct pg_features pg_features;
typedef enum { PG_F_I32 = 1, PG_F_I64 = 2, PG_F_F32 = 3, PG_F_F64 = 4 } pg_feature_dtype;
int pg_features_create(pg_ctx* ctx, uint64_t rows, pg_features** out);
int pg_features_destroy(pg_ctx* ctx, pg_features* fs);
/* add or replace column `name`; host_values: [rows] of the dtype, or NULL (every row = default) */
int pg_features_set_column(pg_ctx* ctx, pg_features* fs, const char* name, int dtype,
                           const void* host_values, double default_value);
int pg_features_column_index(const pg_features* fs, const char* name);   /* -1 if absent */
int pg_features_num_columns(const pg_features* fs);
/* d_out[i][f] = integer column col_idx[f] at d_rows[i] as int32 (int64 saturates) — e.g. the FM field ids */
int pg_features_gather_i32_dev(pg_ctx* ctx, const pg_features* fs, const int32_t* col_idx, uint32_t n_cols,
                               const uint32_t* d_rows, uint32_t n, int32_t* d_out);
/* d_out[i][f] = fmaf((float)value, scale[f], bias[f]) (host arrays [n_cols]; NULL = 1 / 0): dense float
 * inputs with the simplest normalizer fused; richer ones go through pg_expr_* */
int pg_features_gather_f32_dev(pg_ctx* ctx, const pg_features* fs, const int32_t* col_idx, uint32_t n_cols,
                               const float* scale, const float* bias, const uint32_t* d_rows, uint32_t n,
                               float* d_out);

/* d_out[i] = the expression with every variable bound to the feature column of that name at d_rows[i] (the column default for a row
 * outside the store), evaluated in fp64 on the device — the numeric `expression` normalizer of a new_feature over item features
 * (service/feature/new_feature_op.go:54-115 with an ExpressionNormalizer, normalizer.go:112-138; service/feature/feature.go:73-78 walks
 * the items one by one) for a candidate batch, without boxing a map per item.  `e`: pg_expr_compile (pairec's default grammar, `${col}`)
 * or pg_expr_compile_typed(…, "antlr") for the arithmetic subset; at most 16 variables, each of which must be a column (PG_ERR_INVALID
 * names the first that is not); PG_ERR_ARITH as pg_expr_eval_dev. */
int pg_features_eval_dev(pg_ctx* ctx, const pg_features* fs, const pg_expr* e, const uint32_t* d_rows, uint32_t n, double* d_out);

/* A Hologres vector recall with its WhereClause (HologresVectorConf.WhereClause, recconf.go:492-497; hologres_vector_recall.go:
 * 23,49-62 and _v2.go:23,56-61: "FROM table WHERE … ORDER BY distance LIMIT n"), in the shape the device serves: `column OP
 * constant` over an int32 / int64 feature column keyed by item row ("create_time > ${time}" with the constant substituted by the
 * caller).  Only rows that pass are candidates; out_count[q] = min(k, rows that pass), the slots behind it carry row UINT64_MAX.
 * metric 0: inner product, descending (pg_recall_topk); 1: squared Euclidean distance, ascending (pg_recall_topk_l2).  Exact.
 * A filter that admits at most an eighth of the table (and at most 8 M rows; knobs "where_compact_max_rows",
 * "where_compact_min_ratio") is served from a compact copy of the admitted rows, gathered per call; a wider one in place, the
 * predicate evaluated where candidates are made.  0.7-6 ms per call of 1-128 queries at 100 M x 128 for any selectivity. */
typedef enum { PG_WHERE_GT = 0, PG_WHERE_GE = 1, PG_WHERE_LT = 2, PG_WHERE_LE = 3, PG_WHERE_EQ = 4, PG_WHERE_NE = 5 } pg_where_op;
int pg_recall_topk_where(pg_ctx* ctx, const pg_table* t, const pg_features* fs, int column, int op, long long value, int metric,
                         const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows, float* out_scores, uint32_t* out_count);
/* A filtered VIEW of a table: the rows `column OP value` admits, in row order, copied into a table of their own whose recalls answer
 * with the SOURCE's row ids (ties by source row, as pg_recall_topk_where).  For a WhereClause whose constant is fixed when the recall
 * is built (hologres_vector_recall.go:56-61 substitutes "${time}" in the constructor): build the view once per table generation and
 * every recall call — pg_recall_topk[_l2][_dev], pg_coalescer_recall[_l2] / _online_recall of a coalescer created over the view —
 * serves it at the speed of an unfiltered table of that size (its own shadows, statistics and threshold model; requests of many
 * callers share a pass).  A snapshot: later changes of the source or the column do not reach it.  Views serve recall calls only: the
 * recommend calls, and a view as i2i TRIGGER table, are refused (PG_ERR_UNSUPPORTED / PG_ERR_INVALID); rank / DPP / SSD calls take the
 * source table and the recalled ids.  PG_ERR_EMPTY when no row passes (PG_ERR_INVALID is a bad column / operator / feature store).  Destroyed with pg_table_destroy. */
int pg_table_view_create(pg_ctx* ctx, const pg_table* t, const pg_features* fs, int column, int op, long long value, pg_table** out_view);
/* FM + two-tower rank straight from candidate rows: the model's item field ids are the integer columns
 * item_field_cols[n_item_fields] of `fs` (out-of-vocabulary ids are clamped as in pg_rank_fm2t_dev) */
int pg_rank_fm2t_rows_dev(pg_ctx* ctx, const pg_model* m, const pg_features* fs, const int32_t* item_field_cols,
                          const float* d_user_vecs, const int32_t* d_user_field_ids, const uint32_t* d_cand_rows,
                          const uint32_t* d_req_offsets, uint32_t n_req, uint32_t n_items, float* d_out_scores);

/* host-buffer form of the same: the EasyRec flavour of IAlgorithm.Run (service/rank/algo_data.go:79-86: item ids + columnar
 * context features) with the columns already resident — the shim passes candidate rows, nothing is boxed per request */
int pg_rank_fm2t_rows(pg_ctx* ctx, const pg_model* m, const pg_features* fs, const int32_t* item_field_cols,
                      const float* user_vecs, const int32_t* user_field_ids, const uint32_t* cand_rows,
                      const uint32_t* req_offsets, uint32_t n_req, float* out_scores);

/* ---- materialised item records (FM + two-tower) ---------------------------------------------------------------
 * An item's field ids are static per item (the columns of `fs`), so the item side of model `m` can be laid out once, at
 * model-load / column-set time: record r = the embeddings of row r's ids concatenated in field order (128 fp32) followed
 * by their linear weights — 640 B, ONE contiguous gather per candidate at rank time instead of the id row plus
 * n_item_fields scattered 64-B embedding rows (each of which costs a whole 128-B line of HBM traffic).  The records
 * hold the very values the per-field path reads and the kernel sums them in the same order, so scores are bit-identical
 * to pg_rank_fm2t_rows_dev in both precision modes; candidates outside the store read the columns' defaults as there.
 * After a column changes (pg_features_set_column) or the model's field tables are reloaded, _update re-materialises
 * rows [row0, row0 + nrows); the per-field path stays available for field-table hot-swaps. */
typedef struct pg_item_rows pg_item_rows;
int pg_fm2t_item_rows_build(pg_ctx* ctx, const pg_model* m, const pg_features* fs, const int32_t* item_field_cols,
                            pg_item_rows** out);
int pg_fm2t_item_rows_update(pg_ctx* ctx, pg_item_rows* ir, uint64_t row0, uint64_t nrows);
int pg_fm2t_item_rows_destroy(pg_ctx* ctx, pg_item_rows* ir);
/* pg_rank_fm2t_rows_dev / pg_rank_fm2t_rows over the materialised records */
int pg_rank_fm2t_irows_dev(pg_ctx* ctx, const pg_model* m, const pg_item_rows* ir, const float* d_user_vecs,
                           const int32_t* d_user_field_ids, const uint32_t* d_cand_rows, const uint32_t* d_req_offsets,
                           uint32_t n_req, uint32_t n_items, float* d_out_scores);
int pg_rank_fm2t_irows(pg_ctx* ctx, const pg_model* m, const pg_item_rows* ir, const float* user_vecs,
                       const int32_t* user_field_ids, const uint32_t* cand_rows, const uint32_t* req_offsets, uint32_t n_req,
                       float* out_scores);

/* ---- the whole hot path in one call ------------------------------------------------------------
 * One request batch through VectorRecall.GetCandidateItems → RankService.Rank (one DNN3 rank algorithm) →
 * RankScore fusion → ItemRankScoreSort (service/user_recommend.go:83-151 restricted to the hot path),
 * device-resident: recall top-k of `t` for nq user vectors, rank every candidate with `m` (the same user
 * vectors are the model's user features), fuse with `e` — whose variables must be `rank_var` (the model's
 * name in RankAlgoList) and/or "current_score" (Item.Score, i.e. the recall score, module/item.go:189-212) —
 * and sort each request's candidates by the fused score, descending.
 * Outputs, all [nq][k]: global row ids and recall scores in recall order, the model's scores, the fused fp64
 * scores (same order), and d_out_order = positions 0..k-1 of each request sorted by fused score.
 * d_out_count (optional, [nq]) receives each request's number of real candidates: a table with fewer than k rows
 * pads every list with row = UINT64_MAX, recall score = -inf, model score = 0, fused score = NaN — the sort puts
 * those slots last, so the first d_out_count[q] positions of a request's order are its items.
 * The stages are enqueued back to back and verified once, at the end; the call returns after that check. */
int pg_recommend_dnn3_dev(pg_ctx* ctx, const pg_table* t, const pg_model* m, const pg_expr* e, const char* rank_var,
                          const float* d_queries, uint32_t nq, uint32_t k, uint64_t* d_out_rows,
                          float* d_out_recall_scores, float* d_out_rank_scores, double* d_out_fused,
                          uint32_t* d_out_order, uint32_t* d_out_count);

/* The same batch in two halves, for callers that keep several batches queued on the stream (bench.py; a serving loop
 * that owns its batching): _begin enqueues everything and returns at once with a ticket, _end waits for that batch,
 * verifies it (re-running it with the fallback recall plan if the first one did not hold) and frees the ticket.
 * Every ticket must be ended; output buffers belong to the batch until then. */
typedef struct pg_ticket pg_ticket;
int pg_recommend_dnn3_begin(pg_ctx* ctx, const pg_table* t, const pg_model* m, const pg_expr* e, const char* rank_var,
                            const float* d_queries, uint32_t nq, uint32_t k, uint64_t* d_out_rows,
                            float* d_out_recall_scores, float* d_out_rank_scores, double* d_out_fused,
                            uint32_t* d_out_order, uint32_t* d_out_count, pg_ticket** out);
/* scan_ms (optional): the batch's scan-stage launches, HIP-event timed (what pg_last_scan_kernel_ms reports) */
int pg_recommend_end(pg_ctx* ctx, pg_ticket* ticket, double* scan_ms);

/* ---- shard group: one process, several GPUs --------------------------------------------------------
 * BASELINE.json configs[4] / SURVEY.md 8e behind the C ABI (a cgo host cannot join a torch.distributed job): the item
 * table in contiguous row ranges [g*N/G, (g+1)*N/G), one context per shard, model weights replicated.  devices[] may
 * name one device several times (logical shards on one GPU: tests).  Peer access is enabled between distinct
 * devices; the two exchanges of a step (per-shard top-k lists; owners' rank scores and DPP embeddings) are direct
 * peer stores ordered by HIP events — no collective library, no host synchronisation inside a step.  The tail of a
 * step (RankScore, sort, DPPSort, page) is spread over the shards by request: shard s finishes requests q = s (mod G). */
typedef struct pg_group pg_group;
int pg_group_create(const int* devices, uint32_t n_shards, pg_group** out);
int pg_group_destroy(pg_group* g);
uint32_t pg_group_size(const pg_group* g);
int pg_group_info(const pg_group* g, uint64_t* total_rows, uint32_t* dim);
pg_ctx* pg_group_ctx(pg_group* g, uint32_t shard);
pg_table* pg_group_table(pg_group* g, uint32_t shard);
int pg_group_table_create(pg_group* g, uint64_t total_rows, uint32_t dim);
int pg_group_table_fill_synthetic(pg_group* g, uint64_t seed, int normalize);
int pg_group_table_upload(pg_group* g, uint64_t row0, uint64_t nrows, const float* host_rows);   /* global rows */
int pg_group_model_load(pg_group* g, pg_model_kind kind, pg_prec prec, const void* blob, size_t len);
typedef struct {
    uint32_t k;                /* recall depth (RecallCount) */
    uint32_t dpp_candidates;   /* DPPSortConfig.CandidateCount; 0 = no DPP stage (the page is the head of the sorted list) */
    double   dpp_alpha;        /* DPPSortConfig.Alpha */
    uint32_t dpp_window;       /* DPPSortConfig.WindowSize (0 = 10) */
    int      dpp_normalize_emb;
} pg_group_plan;
/* nq requests through sharded recall → merge → owner-computes DNN3 rank → RankScore → ItemRankScore sort → DPPSort on
 * the first max(top_n, dpp_candidates) entries (sort/dpp_sort.go:271-351; ctx.Size = top_n).  Outputs [nq][top_n] in
 * page order: global row ids, recall / model / fused scores; out_count[q] (optional) = entries that are items.
 * Results are identical to the single-shard path (pg_recommend_dnn3_dev + pg_dpp on one table). */
int pg_group_recommend(pg_group* g, const pg_expr* e, const char* rank_var, const pg_group_plan* plan,
                       const float* user_vecs, uint32_t nq, uint32_t top_n, uint64_t* out_rows,
                       float* out_recall_scores, float* out_rank_scores, double* out_fused, uint32_t* out_count);
/* The same step in two halves, as pg_recommend_dnn3_begin / pg_recommend_end on one GPU: _begin enqueues the whole
 * step on every shard and returns at once (user_vecs are copied), _end waits for it, verifies every shard's recall
 * plan (re-running the step where one did not hold) and delivers the pages.  Up to two steps may be outstanding: each
 * shard runs them on two lanes (contexts with their own stream and scratch), so one batch's fusion / sort / DPP tail
 * overlaps the next batch's scans.  Every ticket must be ended; table / model changes need the group idle. */
typedef struct pg_group_ticket pg_group_ticket;
int pg_group_recommend_begin(pg_group* g, const pg_expr* e, const char* rank_var, const pg_group_plan* plan,
                             const float* user_vecs, uint32_t nq, uint32_t top_n, pg_group_ticket** out);
int pg_group_recommend_end(pg_group* g, pg_group_ticket* ticket, uint64_t* out_rows, float* out_recall_scores,
                           float* out_rank_scores, double* out_fused, uint32_t* out_count);
/* The first exchange of the sharded step (the reference's only fan-in is service/recall.go:126-150): every shard
 * sends the best ceil(k/G + 6 sqrt(k/G) + 8) entries of every request's list; a step in which some shard's last sent entry lies
 * inside a merged top-k is repeated with the whole lists.  out4 = {steps served, steps that had to be repeated, bytes one shard
 * sent to each peer in the last step, entries per request and shard in it}. */
int pg_group_exchange_stats(pg_group* g, uint64_t* out4);

/* The shard-side steps of the same flow as device-level calls, for a one-process-per-GPU host that runs the two
 * exchanges itself (pairec_amd/dist.py over torch.distributed / RCCL).  Everything is fixed-size and stays on the
 * stream: no counts travel to the host.
 *   pg_topk_merge_lists_dev   pg_topk_merge_dev with the input layout selectable: list_major = 1 takes
 *                             [nlists][nq][per_list], what an all-gather of per-shard [nq][per_list] blocks produces
 *   pg_owned_compact_dev      of the merged global rows [nq][k], the candidates whose rows live in `t`: their local row
 *                             indices and their slots q * k + j, compacted request by request (stable), and the CSR
 *                             offsets d_req_offsets[nq + 1] — inputs of pg_rank_dnn3_dev (pass n_items = nq * k, an upper bound)
 *   pg_scatter_f32_dev        d_out[d_slot[i]] = d_vals[i] for i < *d_total (= d_req_offsets[nq]); d_out is pre-zeroed by the
 *                             caller, the owners' slots are disjoint, so a sum all-reduce of the slabs is exact
 *   pg_dpp_candidates_dev     global rows and relevance (fused score) of the first n_cand entries of every sorted list
 *   pg_gather_owned_rows_dev  d_out[i][dim] = row d_global_rows[i] where this shard owns it; other rows are left as they are
 *   pg_dpp_batch_dev          DPPSort.KernelMatrix + DPPWithWindow for n_req independent requests of n candidates given as
 *                             embedding rows d_emb [n_req][n][dim]; d_out_idx [n_req][topn], d_out_count [n_req] */
int pg_topk_merge_lists_dev(pg_ctx* ctx, const uint64_t* d_rows, const float* d_scores, uint32_t nq, uint32_t nlists,
                            uint32_t per_list, int list_major, uint32_t k, uint64_t* d_out_rows, float* d_out_scores);
int pg_owned_compact_dev(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t nq, uint32_t k,
                         uint32_t* d_local, uint32_t* d_slot, uint32_t* d_req_offsets);
int pg_scatter_f32_dev(pg_ctx* ctx, const float* d_vals, const uint32_t* d_slot, const uint32_t* d_total, uint32_t cap,
                       float* d_out);
int pg_dpp_candidates_dev(pg_ctx* ctx, const uint32_t* d_order, const uint64_t* d_rows, const double* d_fused, uint32_t nq,
                          uint32_t k, uint32_t n_cand, uint64_t* d_c_rows, double* d_c_rel);
int pg_gather_owned_rows_dev(pg_ctx* ctx, const pg_table* t, const uint64_t* d_global_rows, uint32_t n, float* d_out);
int pg_dpp_batch_dev(pg_ctx* ctx, const float* d_emb, const double* d_rel, uint32_t n_req, uint32_t n, uint32_t dim,
                     double alpha, uint32_t topn, uint32_t window, int normalize_emb, uint32_t* d_out_idx,
                     uint32_t* d_out_count);
/* DPPSort.KernelMatrix alone (sort/dpp_sort.go:372-475, table path): d_out_L [n_req][n][n] = diag(r) F F^T diag(r) in fp64 for the
 * same inputs as pg_dpp_batch_dev — what its greedy part consumes, exposed so that the matrix can be checked bit for bit */
int pg_dpp_kernel_matrix_dev(pg_ctx* ctx, const float* d_emb, const double* d_rel, uint32_t n_req, uint32_t n, uint32_t dim,
                             double alpha, int normalize_emb, double* d_out_L);

/* ---- request coalescer --------------------------------------------------------------------------
 * The reference calls its plug-ins once per request from many goroutines at once: one IAlgorithm.Run per recall
 * (service/recall.go:129-145 → vector_recall.go:88), one per batch of BatchCount = 100 items and per algorithm
 * (service/rank/rank_service.go:264-289), across overlapping HTTP requests (SURVEY.md 8b "Threading").  A table
 * pass costs the same for 1 query as for 128, so the library batches across callers itself: the pg_coalescer_*
 * calls below are issued concurrently from any number of host threads, each with ONE request; they block while a
 * library-owned dispatcher thread forms a batch — up to max_batch requests; a partial batch goes out once the
 * device has nothing queued and its oldest request has waited max_wait_us — runs ONE table pass / ONE rank launch
 * for the whole batch on the context's stream, and hands every caller its slice.  Up to `depth` batches are in
 * flight, so the next batch is queued behind the running one.  Results are bit-identical to the same request
 * issued alone through pg_recall_topk / pg_rank_dnn3 / pg_recommend_dnn3_dev (scores do not depend on what else
 * shares a pass).  cgo note: the calling goroutine's OS thread is parked in a futex wait, not spinning.
 * A coalesced batch records none of its context's stage timers (pg_stats' last_recall_ms / last_rank_ms, pg_last_scan_kernel_ms
 * keep what the last direct call left; the byte count is kept up): each HIP event record is ~6 us of idle queue, 40-60 us of a
 * small batch's 1.3-1.8 ms.  PG_COALESCER_TIMERS=1 in the environment records them as direct calls do; pg_coalescer_stats'
 * device_ms (enqueue -> completion) is there either way. */
typedef struct pg_coalescer pg_coalescer;
typedef struct {
    uint32_t k;               /* recall depth (RecallConfig.RecallCount), fixed per coalescer: 1..16384            */
    uint32_t max_batch;       /* requests per table pass, 1..256 (<= 32 when dim > 128); 0 = the maximum           */
    uint32_t max_wait_us;     /* how long a request may wait for company while the device is idle; 0 = 100.  Only
                               * while company is likely: once the recent gap between arrivals exceeds it, a request
                               * that finds the device idle is dispatched at once                                    */
    uint32_t depth;           /* batches in flight, 1..4; 0 = 2                                                    */
    uint32_t max_top_n;       /* pg_coalescer_recommend: largest page a caller may ask for, <= k; 0 = k            */
    uint32_t max_rank_items;  /* pg_coalescer_rank*: most candidates in one call (BatchCount); 0 = k               */
    uint32_t timeout_us;      /* deadline of every call, from its arrival (eas/client.go:53-58, 100 ms there);
                                 0 = none.  A caller whose deadline passes returns PG_ERR_TIMEOUT; its batch still
                                 completes for the other callers, and later calls are served as usual             */
} pg_coalescer_config;
/* model / expr / rank_var may be NULL: then only pg_coalescer_recall (and pg_coalescer_rank_dnn3 with a model) work. */
int pg_coalescer_create(pg_ctx* ctx, const pg_table* t, const pg_model* m, const pg_expr* e, const char* rank_var,
                        const pg_coalescer_config* cfg, pg_coalescer** out);

/* A scene's whole plug-in set behind one coalescer — what recconf names per scene (RecallConfs, RankConf.RankAlgoList +
 * RankScore, SortNames; recconf/recconf.go:54-57,736-745), so that EVERY per-request plug-in call of the reference has a
 * coalesced single-request entry point below:
 *   algos[]            RankAlgoList: up to 4 rank algorithms, each scoring every candidate (rank_service.go:259-289) —
 *                      PG_MODEL_DNN3 over the table's rows, or PG_MODEL_FM_TWOTOWER whose item field ids are the integer
 *                      columns item_field_cols[n_item_fields] of `features` (keyed by the same rows); `name` is the
 *                      algorithm's name = its variable in RankScore (rank_service.go:312-335)
 *   rank_score         RankConf.RankScore over the algorithms' names and current_score (needed by pg_coalescer_recommend*)
 *   rerank             1: DPPSort behind the ItemRankScore sort (SortNames: [ItemRankScore, DPPSort]): the first
 *                      rerank_candidates = max(ctx.Size, CandidateCount) entries of every sorted list are the DPP
 *                      candidates (sort/dpp_sort.go:280-291), `dpp` carries alpha / window / normalize_emb /
 *                      norm_relevance_score, the page is DPPWithWindow's pick sequence; max_top_n <= rerank_candidates
 *   query_model        OnlineVectorRecall: a PG_MODEL_FM_TWOTOWER whose user tower turns a request's user features into
 *                      the query (pg_coalescer_online_recall); the table is then the item-embedding table (dim = t_out)
 *   trigger_table      I2IVectorRecall: where trigger rows are looked up (NULL = the table itself)
 *   max_rerank_items   pg_coalescer_dpp: most candidates in one call (0 = 1024); max_hook_dim: widest hook embedding (0 = none) */
typedef struct {
    const pg_model* model;
    const char* name;
    const pg_features* features;
    const int32_t* item_field_cols;
    const pg_item_rows* item_rows;   /* optional: the materialised records of (model, features, columns) — preferred when set */
    /* a multi-output model (PG_MODEL_DNN3_MULTI): the names of its pg_model_num_outputs() outputs; RankScore then reads
     * "<name>_<output>" per output (rank_service.go:315-319), pg_coalescer_recommend_ex returns one rank plane per output
     * (in list order, an algorithm's outputs adjacent), pg_coalescer_rank / _rank_dnn3 write out_scores[o * n + i].
     * NULL for single-output models; NULL for a multi-output one names its outputs "0", "1", … */
    const char* const* output_names;
} pg_rank_algo;
typedef struct {
    pg_coalescer_config base;
    const pg_rank_algo* algos;
    uint32_t n_algos;
    const pg_expr* rank_score;
    int rerank;
    uint32_t rerank_candidates;
    pg_dpp_options dpp;
    const pg_model* query_model;
    const pg_table* trigger_table;
    uint32_t max_rerank_items;
    uint32_t max_hook_dim;
} pg_scene_config;
int pg_coalescer_create_scene(pg_ctx* ctx, const pg_table* t, const pg_scene_config* cfg, pg_coalescer** out);
/* fails every waiting request with PG_ERR_INVALID, joins the worker threads, frees the buffers */
int pg_coalescer_destroy(pg_coalescer* c);
/* VectorRecall.GetCandidateItems → IAlgorithm.Run(VectorRequest{K, Vector}) for ONE user vector [dim]:
 * out_rows[k], out_scores[k] as pg_recall_topk, *out_count (optional) = valid entries. */
int pg_coalescer_recall(pg_coalescer* c, const float* query, uint64_t* out_rows, float* out_scores,
                        uint32_t* out_count);
/* I2IVectorRecall.GetCandidateItems for ONE trigger item (pg_i2i_recall with n = 1): the query is row `trigger_row` of
 * the scene's trigger table; rides the same table pass as the vector recalls of other callers. */
/* HologresVectorRecallV2 through the coalescer: one request per call, up to 128 share one screened pass (32 one pass of the
 * exact scan, where the table needs that: pg_recall_topk_l2); out_dist ascending. */
int pg_coalescer_recall_l2(pg_coalescer* c, const float* query, uint64_t* out_rows, float* out_dist, uint32_t* out_count);
int pg_coalescer_i2i_recall(pg_coalescer* c, uint32_t trigger_row, uint64_t* out_rows, float* out_scores,
                            uint32_t* out_count);
/* OnlineVectorRecall.GetCandidateItems for ONE user (pg_online_vector_recall with n_req = 1): user_vec[d_user] goes
 * through the scene's query_model user tower on the device, the embedding is the query. */
int pg_coalescer_online_recall(pg_coalescer* c, const float* user_vec, uint64_t* out_rows, float* out_scores,
                               uint32_t* out_count);
/* IAlgorithm.Run of rank algorithm `algo` (index into the scene's list) for ONE batch of n <= max_rank_items
 * candidates of one user (rank_service.go:273): cand_rows are local row indices, out_scores[n] in request order.
 * user_vec is [d_user] of that model; user_field_ids [n_user_fields] for an FM + two-tower model, else NULL.
 * Calls for one algorithm from any number of threads and requests become one launch. */
int pg_coalescer_rank(pg_coalescer* c, uint32_t algo, const float* user_vec, const int32_t* user_field_ids,
                      const uint32_t* cand_rows, uint32_t n, float* out_scores);
/* the same for the first DNN3 / the first FM + two-tower algorithm of the scene */
int pg_coalescer_rank_dnn3(pg_coalescer* c, const float* user_vec, const uint32_t* cand_rows, uint32_t n,
                           float* out_scores);
int pg_coalescer_rank_fm2t(pg_coalescer* c, const float* user_vec, const int32_t* user_field_ids,
                           const uint32_t* cand_rows, uint32_t n, float* out_scores);
/* The whole path for ONE request (what pg_recommend_dnn3_dev does for a batch): recall top-k → every rank algorithm →
 * RankScore → descending sort → (DPPSort when the scene has the stage); the caller receives the page of top_n <=
 * max_top_n entries — the head of the sorted list, or DPP's picks in pick order — as global row ids, recall scores,
 * model scores and fused scores, all [top_n], and *out_count (optional) = entries that are items.
 * pg_coalescer_recommend reports the first algorithm's scores; _ex takes the FM models' user field ids and reports
 * every algorithm's scores, out_rank_scores [n_algos][top_n]. */
int pg_coalescer_recommend(pg_coalescer* c, const float* user_vec, uint32_t top_n, uint64_t* out_rows,
                           float* out_recall_scores, float* out_rank_scores, double* out_fused,
                           uint32_t* out_count);
int pg_coalescer_recommend_ex(pg_coalescer* c, const float* user_vec, const int32_t* user_field_ids, uint32_t top_n,
                              uint64_t* out_rows, float* out_recall_scores, float* out_rank_scores, double* out_fused,
                              uint32_t* out_count);
/* DPPSort.Sort for ONE request (pg_dpp_ex; sort/sort.go:65-125 calls it once per request, requests overlap): calls
 * with the same candidate count and options share one batched launch (KernelMatrix for all of them, one wave per
 * request for the greedy part).  has_table = 1: embeddings are rows cand_rows[n] of the table; hook_emb as pg_dpp_ex. */
int pg_coalescer_dpp(pg_coalescer* c, const uint32_t* cand_rows, const double* rel, uint32_t n,
                     const pg_dpp_options* opt, const double* hook_emb, uint32_t* out_idx, uint32_t* out_count,
                     double* out_relevance);
/* ---- per-request calls over several GPUs ------------------------------------------------------------------------
 * (a) The sharded table (BASELINE.json configs[4]): a coalescer over a shard group — pg_coalescer_recommend takes ONE
 *     request, the library batches up to 256 of them into pg_group_recommend_begin / _end steps (two in flight, one per
 *     lane), DPPSort included when plan->dpp_candidates > 0.  Only pg_coalescer_recommend / _stats / _destroy apply.
 * (b) Replicas (the 51 GB table of configs[1] fits every GPU): a router over one coalescer per replica; each request
 *     goes to the replica with the fewest requests outstanding (ties: round robin), so per-request calls from one
 *     process reach all N GPUs with no data-path exchange at all.  The router does not own the coalescers. */
int pg_coalescer_create_group(pg_group* g, const pg_expr* e, const char* rank_var, const pg_group_plan* plan,
                              const pg_coalescer_config* cfg, pg_coalescer** out);
typedef struct pg_router pg_router;
int pg_router_create(pg_coalescer* const* replicas, uint32_t n, pg_router** out);
int pg_router_destroy(pg_router* r);
int pg_router_recommend(pg_router* r, const float* user_vec, uint32_t top_n, uint64_t* out_rows,
                        float* out_recall_scores, float* out_rank_scores, double* out_fused, uint32_t* out_count);
int pg_router_recall(pg_router* r, const float* query, uint64_t* out_rows, float* out_scores, uint32_t* out_count);
/* requests each replica has served so far, [n] */
int pg_router_stats(pg_router* r, uint64_t* out_served);

/* SSDSort.Sort for ONE request (pg_ssd): calls of equal shape share one launch — every request its own set of
 * single-wave workgroups with its own barrier.  Shapes: dim 64 / 128, window <= 16, n <= max_rerank_items (others:
 * PG_ERR_UNSUPPORTED, use pg_ssd).  Arguments and outputs as pg_ssd. */
int pg_coalescer_ssd(pg_coalescer* c, const uint32_t* cand_rows, const double* rel, uint32_t n, double gamma, uint32_t topn,
                     uint32_t window, int normalize_emb, int ensure_pos_similarity, int norm_quality_score, int use_ssd_star,
                     uint32_t* out_idx, uint32_t* out_count, double* out_quality);
typedef struct {
    uint64_t requests[6], batches[6];   /* per flavour: 0 recall (vector / i2i / online), 1 rank, 2 recommend, 3 dpp, 4 ssd, 5 reserved */
    uint64_t largest_batch[6];
    uint64_t replans;                   /* batches whose first recall plan did not hold and was re-run */
    uint64_t timeouts;                  /* calls that returned PG_ERR_TIMEOUT */
    double   device_ms[6];              /* summed enqueue → completion time of the batches */
} pg_coalescer_stats_t;
int pg_coalescer_stats(pg_coalescer* c, pg_coalescer_stats_t* out);

/* Deadline form of pg_recommend_end: waits at most timeout_us for the batch; PG_ERR_TIMEOUT leaves the ticket valid
 * (end it again later — every ticket must be ended). */
int pg_recommend_end_timed(pg_ctx* ctx, pg_ticket* ticket, uint32_t timeout_us, double* scan_ms);
/* Test aid: occupies the context's stream for `ms` milliseconds with a kernel that only watches the clock — what a
 * hung or slow device looks like to callers with a deadline. */
int pg_debug_stall(pg_ctx* ctx, uint32_t ms);

/* ---- stats ----------------------------------------------------------------------------------*/
typedef struct {
    uint64_t recall_calls, recall_rows_scanned, recall_rescans;
    uint64_t rank_calls, rank_items;
    uint64_t sort_calls, sort_items;
    double   last_recall_ms, last_rank_ms, last_sort_ms;   /* hipEvent-timed, device side */
    uint64_t recall_predicted;          /* batches whose screening threshold came from the table's threshold model and held */
    /* screened recalls whose first plan held: the (row, query) pairs the full pass handed to the exact fp32 re-scoring, and the
     * queries they belong to (suspects per answer = recall_suspects / recall_suspect_queries / K); the pairs a mid-batch pass's
     * 4-bit stage handed to its int8 stage; screened plans that overflowed (rows crowded within the screen's error of the K-th
     * score) and finished on the exact scan */
    uint64_t recall_suspects, recall_suspect_queries, recall_i4m_pairs, recall_screen_overflows;
    uint64_t recall_record_growths;     /* batches re-run with larger hit-record areas (the table keeps them) */
    uint64_t recall_rescored;           /* of recall_suspects, the pairs that reached the exact fp32 re-scoring (fewer where a table's
                                           crowded rows switched the two-digit refinement stage on, csrc/recall_r2.hip) */
    uint64_t sort_split_calls;          /* score sorts and top-K final orders that took the split sort (few lists of 1025 … 8192
                                           items: runs sorted wave by wave over the chip, csrc/split_sort.hpp) */
} pg_stats_t;
int pg_stats(pg_ctx* ctx, pg_stats_t* out);
/* time (ms) of the dominant kernel of the last pg_recall_* call, measured with HIP events on the
 * context's stream around the scan launches only (bench.py's roofline figure) */
int pg_last_scan_kernel_ms(pg_ctx* ctx, double* out_ms, uint64_t* out_bytes);
/* Measured HBM read ceiling: streams the table's rows through a plain read-only kernel `reps` times
 * and returns the best rate in GB/s.  SURVEY.md 8(d) asks for the roofline fraction against a measured
 * streaming ceiling beside the nominal 8 TB/s. */
int pg_hbm_read_probe(pg_ctx* ctx, const pg_table* t, int reps, double* out_gbps);
/* The shadow the screened recall streams for this table, built now if it is not yet (it is otherwise built by
 * the first recall after an upload): *out_elem_bytes = 1 (int8: dim 128 with a value range one scale can serve),
 * 2 (bf16: dim 64, and heavy-tailed dim-128 tables), 0 when the table
 * is scanned exactly in fp32 (other dims, non-finite rows, no memory).  int8: *out_scale = the table's
 * quantisation step s (x ~ s X, X in [-127,127]), *out_resid = max over rows of ||x - s X||_2 as measured —
 * the two table-side terms of the screen's error bound; 0 otherwise.  Any out pointer may be NULL. */
int pg_table_screen_info(pg_ctx* ctx, const pg_table* t, int* out_elem_bytes, float* out_scale, float* out_resid);

#ifdef __cplusplus
}
#endif
#endif /* PAIREC_GPU_H */
