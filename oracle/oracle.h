/*
 * oracle.h — CPU restatement of pairec's rank+recall hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libpairec_gpu.so) never links, loads or calls anything in oracle/.
 *
 * PARITY STATUS:
 *   - score fusion expressions, sort order, response widening: pinned against the reference's own
 *     known-answer tests (tests/golden/reference_known_answers.json).
 *   - recall top-K, DNN / FM predict, DPP numerics: **parity unpinned** — in the reference that
 *     arithmetic runs in remote services (FAISS / PAI-EAS) or in un-vendored gonum v0.12.0, and
 *     the reference has no tests for it.  This file *is* the specification for those stages.
 */
#ifndef PAIREC_ORACLE_H
#define PAIREC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic data ------------------------------------------------------- */
uint64_t orc_splitmix64(uint64_t x);
float    orc_synth_value(uint64_t seed, uint64_t row, uint32_t col, uint32_t dim);
/* rows [row0,row0+nrows) of the synthetic table; normalize!=0 → L2-normalised in fp32 */
void orc_synth_rows(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, int normalize,
                    float* out);
void orc_synth_mixture_rows(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, uint32_t n_centres, float noise_scale,
                            uint64_t stream, float* out);
/* uniform [-scale,scale) vector of n values: value i = synth_value(seed, 0, i, n) * scale */
void orc_synth_uniform(uint64_t seed, uint64_t n, float scale, float* out);

/* ---- recall: inner-product top-K (vector_recall.go:70-102, hologres_vector_recall.go:23) --- */
uint64_t orc_topk_key(float score, uint32_t row);
void orc_dot_scores(const float* table, uint64_t nrows, uint32_t dim, const float* queries,
                    uint32_t nq, float* out /* [nq][nrows] */, int threads);
/* exact top-K (score desc by IEEE total order, row asc) of rows [0,nrows); out_rows are
 * row_offset + local row.  Returns number written per query (min(K,nrows)). */
uint32_t orc_recall_topk(const float* table, uint64_t nrows, uint32_t dim, uint64_t row_offset,
                         const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows,
                         float* out_scores, int threads);
/* squared Euclidean distance, ascending (hologres_vector_recall_v2.go:23): d = fmaf(-2, ip, |x|^2 + |q|^2), chains k-ascending */
uint32_t orc_recall_topk_l2(const float* table, uint64_t nrows, uint32_t dim, uint64_t row_offset,
                            const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows,
                            float* out_dist, int threads);
/* merge G per-shard top-K lists (keys = score,row) into the global top-K */
uint32_t orc_topk_merge(const uint64_t* rows, const float* scores, uint32_t nlists,
                        uint32_t per_list, uint32_t k, uint64_t* out_rows, float* out_scores);

/* ---- rank: 3-layer DNN (algorithm/eas-shaped predict) ------------------------------------- */
typedef struct {
    uint32_t d_user, d_item, h1, h2;   /* 128,128,512,256 for cfg 3 */
    const float* w1;  /* [(d_user+d_item)][h1] row-major (k major) */
    const float* b1;  /* [h1] */
    const float* w2;  /* [h1][h2] */
    const float* b2;  /* [h2] */
    const float* w3;  /* [h2] */
    float b3;
} orc_dnn3;
/* prec: 0 = f32 (parity mode, bit-defined chains), 1 = bf16 operands / f32 accumulate */
void orc_dnn3_forward(const orc_dnn3* m, int prec, const float* user_vec,
                      const float* item_rows /* [n][d_item] */, uint64_t n, float* out_scores,
                      int threads);
uint16_t orc_f32_to_bf16(float x);
float    orc_bf16_to_f32(uint16_t x);

/* ---- rank: FM + two-tower (cfg 4) ---------------------------------------------------------- */
typedef struct {
    uint32_t n_user_fields, n_item_fields, k;      /* 8, 8, 16 */
    uint32_t d_user, t_h1, t_out;                  /* user tower 128 -> 256 -> 64 */
    const float* fm_w;      /* per-field linear weight tables are folded into embeddings:     */
    float fm_b;             /* y_lin = fm_b + sum_f lin[f][id]  (lin tables passed per call)  */
    const float* uw1; const float* ub1; const float* uw2; const float* ub2; /* user tower  */
    const float* iw1; const float* ib1; const float* iw2; const float* ib2; /* item tower  */
} orc_fm2t;
/* field_emb: [n_fields] pointers to [vocab][k] tables; field_lin: [n_fields] pointers to [vocab] */
void orc_fm2t_user_embedding(const orc_fm2t* m, int prec, const float* user_vec, float* uo);
void orc_fm2t_forward(const orc_fm2t* m, int prec, const float* const* field_emb,
                      const float* const* field_lin, const float* user_vec,
                      const int32_t* user_field_ids, const int32_t* item_field_ids /* [n][nif] */,
                      uint64_t n, float* out_scores, int threads);

/* ---- sort (sort/item_score.go:15-18, sort/item_rank_score.go:26-32) ------------------------ */
/* order by f64 score; desc!=0 → ItemRankScoreSort, else ItemScoreSort.  Ties: input index asc.
 * NaN sorts last in both directions (reference: undefined — Go's sort.Sort with a < comparator). */
void orc_sort_scores(const double* scores, uint32_t n, int desc, uint32_t* out_order);

/* ---- DPP (sort/dpp_sort.go:372-551) -------------------------------------------------------- */
/* emb: [n][d] fp64 (already normalised when normalize_emb was applied by the caller);
 * rel: relevance scores; returns selected indices (count = min(topn, n)). */
void orc_dpp_kernel_matrix(const double* emb, uint32_t n, uint32_t d, const double* rel,
                           double alpha, double* L /* [n][n] */);
void orc_dpp_kernel_matrix_f(const double* F, uint32_t n, uint32_t d1, const double* rel, double alpha, double* L);
uint32_t orc_dpp_with_window(const double* L, uint32_t n, uint32_t topn, uint32_t window,
                             uint32_t* out_idx);

/* ---- SSD (sort/ssd_sort.go:346-486) -------------------------------------------------------- */
int orc_ssd_quality(const double* rel, uint32_t n, int mode, double* out);
uint32_t orc_ssd_window(double* emb, uint32_t n, uint32_t d, const double* rel, double gamma,
                        uint32_t topn, uint32_t window, int use_ssd_star, uint32_t* out_idx);
void orc_l2_normalize_f64(double* v, uint32_t d);

#ifdef __cplusplus
}
#endif
#endif
