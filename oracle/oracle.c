/*
 * oracle.c — CPU restatement of pairec's rank+recall hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h for who may load this and for the parity status of each stage).
 *
 * Every numeric stage below fixes a *summation order* so that the HIP kernels can be compared
 * bit-for-bit where the arithmetic allows it:
 *   chain(c; a_k*b_k, k asc)  :=  acc=c; for k: acc = fmaf(a_k, b_k, acc)      (one rounding/step)
 * which is exactly what gfx950's f32-input MFMA computes (k-ordered fmaf chain).
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include "oracle.h"
#include <float.h>
#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* synthetic data: SURVEY.md §8(d).  u = splitmix64(seed ^ (row*D + col));                      */
/* value = (u>>40) * 2^-24 * 2 - 1  (uniform fp32 in [-1,1)); rows L2-normalised in fp32.       */
/* ------------------------------------------------------------------------------------------ */
uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

float orc_synth_value(uint64_t seed, uint64_t row, uint32_t col, uint32_t dim) {
    uint64_t u = orc_splitmix64(seed ^ (row * (uint64_t)dim + col));
    /* (u>>40) < 2^24 is exact in fp32; *2^-23 exact; -1 exact (result is a multiple of 2^-23) */
    return (float)(u >> 40) * (1.0f / 8388608.0f) - 1.0f;
}

void orc_synth_rows(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, int normalize,
                    float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)nrows; ++r) {
        float* o = out + (size_t)r * dim;
        float ss = 0.0f;
        for (uint32_t c = 0; c < dim; ++c) {
            float v = orc_synth_value(seed, row0 + (uint64_t)r, c, dim);
            o[c] = v;
            ss = fmaf(v, v, ss);                 /* chain over c asc */
        }
        if (normalize) {
            float inv = 1.0f / sqrtf(ss);        /* IEEE sqrt then IEEE divide */
            for (uint32_t c = 0; c < dim; ++c) o[c] = o[c] * inv;
        }
    }
}

/* clustered rows (pairec_amd/csrc/table.hip, table_fill_mixture_kernel): centre c = splitmix64((seed + 2 + 16 stream) ^ g) mod
 * n_centres, x = centre (normalised synthetic row c of seed + 1) + noise_scale * uniform(seed + 3 + 16 stream), normalised.
 * stream 0 is what the device fills a table with; other streams draw further points of the same mixture (queries). */
void orc_synth_mixture_rows(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, uint32_t n_centres, float noise_scale,
                            uint64_t stream, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)nrows; ++r) {
        const uint64_t g = row0 + (uint64_t)r;
        const uint32_t c = (uint32_t)(orc_splitmix64((seed + 2 + 16 * stream) ^ g) % n_centres);
        float* o = out + (size_t)r * dim;
        float ss = 0.0f;
        for (uint32_t k = 0; k < dim; ++k) {
            float v = orc_synth_value(seed + 1, c, k, dim);
            ss = fmaf(v, v, ss);
        }
        const float ic = 1.0f / sqrtf(ss);
        float sx = 0.0f;
        for (uint32_t k = 0; k < dim; ++k) {
            float x = fmaf(orc_synth_value(seed + 3 + 16 * stream, g, k, dim), noise_scale, orc_synth_value(seed + 1, c, k, dim) * ic);
            o[k] = x;
            sx = fmaf(x, x, sx);
        }
        const float ix = 1.0f / sqrtf(sx);
        for (uint32_t k = 0; k < dim; ++k) o[k] = o[k] * ix;
    }
}

void orc_synth_uniform(uint64_t seed, uint64_t n, float scale, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t u = orc_splitmix64(seed ^ i);
        out[i] = ((float)(u >> 40) * (1.0f / 8388608.0f) - 1.0f) * scale;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* recall: score = <row, query> (inner product, descending — hologres_vector_recall.go:23),     */
/* fp32 query/table/score (vectorretrieval.proto:11-20), widened to f64 by the caller           */
/* (vector_recall.go:98).  Order: IEEE-754 totalOrder on the score, then row ascending.         */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t f32_ordered_bits(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    if (f != f) return 0u;                       /* NaN sorts below everything */
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

uint64_t orc_topk_key(float score, uint32_t row) {
    return ((uint64_t)f32_ordered_bits(score) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}

void orc_dot_scores(const float* table, uint64_t nrows, uint32_t dim, const float* queries,
                    uint32_t nq, float* out, int threads) {
    /* transpose queries to [dim][nq] so the q loop vectorises; each (row,q) is still its own
     * k-ascending fmaf chain. */
    float* qt = (float*)malloc((size_t)dim * nq * sizeof(float));
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t k = 0; k < dim; ++k) qt[(size_t)k * nq + q] = queries[(size_t)q * dim + k];
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        float* acc = (float*)malloc(nq * sizeof(float));
#pragma omp for schedule(static)
        for (int64_t r = 0; r < (int64_t)nrows; ++r) {
            const float* x = table + (size_t)r * dim;
            for (uint32_t q = 0; q < nq; ++q) acc[q] = 0.0f;
            for (uint32_t k = 0; k < dim; ++k) {
                const float xv = x[k];
                const float* qk = qt + (size_t)k * nq;
                for (uint32_t q = 0; q < nq; ++q) acc[q] = fmaf(xv, qk[q], acc[q]);
            }
            for (uint32_t q = 0; q < nq; ++q) out[(size_t)q * nrows + r] = acc[q];
        }
        free(acc);
    }
    free(qt);
}

static int cmp_key_desc(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x < y) - (x > y);
}

/* min-heap of keys, size k */
static void heap_sift_down(uint64_t* h, uint32_t n, uint32_t i) {
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l] < h[m]) m = l;
        if (r < n && h[r] < h[m]) m = r;
        if (m == i) return;
        uint64_t t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

static void emit_sorted(uint64_t* keys, uint32_t n, uint64_t row_offset, uint64_t* out_rows,
                        float* out_scores, const float* score_of_local /* may be NULL */) {
    (void)score_of_local;
    qsort(keys, n, sizeof(uint64_t), cmp_key_desc);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t ob = (uint32_t)(keys[i] >> 32);
        uint32_t row = 0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFu);
        uint32_t b = (ob & 0x80000000u) ? (ob & 0x7FFFFFFFu) : ~ob;
        float s;
        memcpy(&s, &b, 4);
        if (ob == 0u) s = NAN;
        out_rows[i] = row_offset + row;
        out_scores[i] = s;
    }
}

/* metric 0: inner product, descending (service/recall/hologres_vector_recall.go:23, pm_approx_inner_product_distance … ORDER BY
 * distance desc).  metric 1: squared Euclidean distance, ascending (service/recall/hologres_vector_recall_v2.go:23,
 * pm_approx_squared_euclidean_distance … ORDER BY distance): specified as d = fmaf(-2, ip, nx + nq) with ip, nx = |x|^2 and
 * nq = |q|^2 each a k-ascending fp32 fmaf chain; rows are ranked by -d = fmaf(2, ip, -(nx + nq)) (exactly the negation), ties by
 * row ascending, and the distance is what comes out as the item's score (hologres_vector_recall_v2.go:181-189). */
static uint32_t recall_topk_metric(const float* table, uint64_t nrows, uint32_t dim, uint64_t row_offset,
                                   const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows,
                                   float* out_scores, int threads, int metric) {
    /* The reference-shaped CPU path: each worker scans a contiguous row range,
     * scores a row against all queries (k-ascending fmaf chains, vectorised across queries) and
     * keeps one min-heap of K keys per query; the per-worker heaps are merged at the end. */
    uint32_t kk = (nrows < k) ? (uint32_t)nrows : k;
    if (kk == 0) return 0;
    float* qt = (float*)malloc((size_t)dim * nq * sizeof(float));
    float* qn = (float*)calloc(nq, sizeof(float));
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t c = 0; c < dim; ++c) {
            qt[(size_t)c * nq + q] = queries[(size_t)q * dim + c];
            qn[q] = fmaf(queries[(size_t)q * dim + c], queries[(size_t)q * dim + c], qn[q]);
        }
    int nth = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    nth = omp_get_max_threads();
#endif
    if ((uint64_t)nth > nrows) nth = (int)nrows;
    uint64_t* heaps = (uint64_t*)malloc((size_t)nth * nq * kk * sizeof(uint64_t));
    uint32_t* hn = (uint32_t*)calloc((size_t)nth * nq, sizeof(uint32_t));
#pragma omp parallel num_threads(nth)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        const uint64_t r0 = nrows * (uint64_t)tid / nth, r1 = nrows * (uint64_t)(tid + 1) / nth;
        /* four rows per sweep over the transposed queries (each (row, query) chain is still its own k-ascending
         * fmaf chain — same bits — but the 4 x nq accumulators stay in L1 and the queries are read once per four rows) */
        enum { RB = 4 };
        float* accb = (float*)malloc((size_t)RB * nq * sizeof(float));
        uint64_t* myh = heaps + (size_t)tid * nq * kk;
        uint32_t* myn = hn + (size_t)tid * nq;
        for (uint64_t rb = r0; rb < r1; rb += RB) {
            const uint32_t nr = (uint32_t)((r1 - rb) < RB ? (r1 - rb) : RB);
            const float* x0 = table + (size_t)rb * dim;
            const float* x1 = table + (size_t)(rb + (nr > 1 ? 1 : 0)) * dim;
            const float* x2 = table + (size_t)(rb + (nr > 2 ? 2 : 0)) * dim;
            const float* x3 = table + (size_t)(rb + (nr > 3 ? 3 : 0)) * dim;
            float* a0 = accb; float* a1 = accb + nq; float* a2 = accb + 2 * (size_t)nq; float* a3 = accb + 3 * (size_t)nq;
            for (uint32_t q = 0; q < nq; ++q) { a0[q] = 0.0f; a1[q] = 0.0f; a2[q] = 0.0f; a3[q] = 0.0f; }
            for (uint32_t c = 0; c < dim; ++c) {
                const float v0 = x0[c], v1 = x1[c], v2 = x2[c], v3 = x3[c];
                const float* qk = qt + (size_t)c * nq;
                for (uint32_t q = 0; q < nq; ++q) {
                    const float qv = qk[q];
                    a0[q] = fmaf(v0, qv, a0[q]);
                    a1[q] = fmaf(v1, qv, a1[q]);
                    a2[q] = fmaf(v2, qv, a2[q]);
                    a3[q] = fmaf(v3, qv, a3[q]);
                }
            }
          for (uint32_t ri = 0; ri < nr; ++ri) {
            const uint64_t r = rb + ri;
            float* acc = accb + (size_t)ri * nq;
            if (metric == 1) {
                const float* x = table + (size_t)r * dim;
                float nx = 0.0f;
                for (uint32_t c = 0; c < dim; ++c) nx = fmaf(x[c], x[c], nx);
                for (uint32_t q = 0; q < nq; ++q) acc[q] = fmaf(2.0f, acc[q], -(nx + qn[q]));
            }
            for (uint32_t q = 0; q < nq; ++q) {
                uint64_t* h = myh + (size_t)q * kk;
                const uint64_t key = orc_topk_key(acc[q], (uint32_t)r);
                uint32_t n = myn[q];
                if (n < kk) {
                    h[n++] = key;
                    myn[q] = n;
                    if (n == kk)
                        for (int32_t i = (int32_t)kk / 2 - 1; i >= 0; --i) heap_sift_down(h, kk, (uint32_t)i);
                } else if (key > h[0]) {
                    h[0] = key;
                    heap_sift_down(h, kk, 0);
                }
            }
          }
        }
        free(accb);
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t q = 0; q < (int32_t)nq; ++q) {
        uint64_t* all = (uint64_t*)malloc((size_t)nth * kk * sizeof(uint64_t));
        uint32_t n = 0;
        for (int t = 0; t < nth; ++t) {
            const uint32_t c = hn[(size_t)t * nq + q];
            memcpy(all + n, heaps + ((size_t)t * nq + q) * kk, (size_t)c * sizeof(uint64_t));
            n += c;
        }
        qsort(all, n, sizeof(uint64_t), cmp_key_desc);
        emit_sorted(all, kk, row_offset, out_rows + (size_t)q * k, out_scores + (size_t)q * k, NULL);
        if (metric == 1)
            for (uint32_t i = 0; i < kk; ++i) out_scores[(size_t)q * k + i] = 0.0f - out_scores[(size_t)q * k + i];   /* (a zero distance comes out as +0) */
        free(all);
    }
    free(hn); free(heaps); free(qt); free(qn);
    return kk;
}

uint32_t orc_recall_topk(const float* table, uint64_t nrows, uint32_t dim, uint64_t row_offset,
                         const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows,
                         float* out_scores, int threads) {
    return recall_topk_metric(table, nrows, dim, row_offset, queries, nq, k, out_rows, out_scores, threads, 0);
}

uint32_t orc_recall_topk_l2(const float* table, uint64_t nrows, uint32_t dim, uint64_t row_offset,
                            const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows,
                            float* out_dist, int threads) {
    return recall_topk_metric(table, nrows, dim, row_offset, queries, nq, k, out_rows, out_dist, threads, 1);
}

uint32_t orc_topk_merge(const uint64_t* rows, const float* scores, uint32_t nlists,
                        uint32_t per_list, uint32_t k, uint64_t* out_rows, float* out_scores) {
    uint32_t n = nlists * per_list;
    uint64_t* keys = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
    for (uint32_t i = 0; i < n; ++i) keys[i] = orc_topk_key(scores[i], (uint32_t)rows[i]);
    qsort(keys, n, sizeof(uint64_t), cmp_key_desc);
    uint32_t kk = n < k ? n : k;
    emit_sorted(keys, kk, 0, out_rows, out_scores, NULL);
    free(keys);
    return kk;
}

/* ------------------------------------------------------------------------------------------ */
/* bf16 helpers: round-to-nearest-even from fp32 (integer formula; NaN stays NaN)              */
/* ------------------------------------------------------------------------------------------ */
uint16_t orc_f32_to_bf16(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((b >> 16) | 0x0040u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}
float orc_bf16_to_f32(uint16_t x) {
    uint32_t b = (uint32_t)x << 16;
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static inline float rbf(float x) { return orc_bf16_to_f32(orc_f32_to_bf16(x)); }
static inline float op_round(float x, int prec) { return prec ? rbf(x) : x; }

static inline float sigmoidf_spec(float z) { return 1.0f / (1.0f + expf(-z)); }

/* ------------------------------------------------------------------------------------------ */
/* generic fused 2-layer MLP + dot head (the shape of both the DNN3 rank model and the item   */
/* tower of the two-tower model):                                                              */
/*   z1_j = chain(c1_j; x_k * W1[k][j], k asc)         h1 = P(relu(z1))                        */
/*   z2_m = chain(b2_m; h1_j * W2[j][m], j asc)        h2 = act2 ? relu(z2) : z2  (fp32, kept) */
/*   z3   = chain(bias3; h2_m*w3_m, m < H2/2) + chain(0; h2_m*w3_m, m >= H2/2)                 */
/*   out  = 1/(1+expf(-z3))                                                                    */
/* P() = bf16 rounding in prec 1 (operands x, W1, W2 are then bf16-rounded too), identity in   */
/* prec 0.  In prec 1 the device accumulates z1/z2 in fp32 inside the bf16 MFMA in an          */
/* unspecified order, so prec 1 is compared with a tolerance; prec 0 is bit-defined.           */
/* ------------------------------------------------------------------------------------------ */
static void mlp2_dot_row(int prec, uint32_t din, uint32_t h1n, uint32_t h2n, const float* x,
                         const float* c1, const float* w1 /* [din][h1n], pre-rounded */,
                         const float* w2 /* [h1n][h2n], pre-rounded */, const float* b2, int act2,
                         const float* w3, float bias3, float* h1buf, float* h2buf, float* out) {
    for (uint32_t j = 0; j < h1n; ++j) h1buf[j] = c1[j];
    for (uint32_t k = 0; k < din; ++k) {
        const float xv = op_round(x[k], prec);
        const float* wr = w1 + (size_t)k * h1n;
        for (uint32_t j = 0; j < h1n; ++j) h1buf[j] = fmaf(xv, wr[j], h1buf[j]);
    }
    for (uint32_t j = 0; j < h1n; ++j) h1buf[j] = op_round(h1buf[j] > 0.0f ? h1buf[j] : 0.0f, prec);
    for (uint32_t m = 0; m < h2n; ++m) h2buf[m] = b2[m];
    for (uint32_t j = 0; j < h1n; ++j) {
        const float hv = h1buf[j];
        const float* wr = w2 + (size_t)j * h2n;
        for (uint32_t m = 0; m < h2n; ++m) h2buf[m] = fmaf(hv, wr[m], h2buf[m]);
    }
    if (act2)
        for (uint32_t m = 0; m < h2n; ++m) h2buf[m] = h2buf[m] > 0.0f ? h2buf[m] : 0.0f;
    float p0 = bias3, p1 = 0.0f;
    const uint32_t half = h2n / 2;
    for (uint32_t m = 0; m < half; ++m) p0 = fmaf(h2buf[m], w3[m], p0);
    for (uint32_t m = half; m < h2n; ++m) p1 = fmaf(h2buf[m], w3[m], p1);
    *out = sigmoidf_spec(p0 + p1);
}

/* The same arithmetic for ORC_MB items at a time: every output element is still its own k-ascending fmaf chain, so
 * the bits are those of mlp2_dot_row — but a row of W1 / W2 is loaded once per ORC_MB items instead of once per item,
 * and the accumulators of a 16-column tile stay in registers (the row-wise form streams the 512-float hidden buffer
 * through L1 for every input element: that is what made the CPU baseline's rank leg a 0.3 TFLOP/s scalar chain). */
#define ORC_MB 4
#define ORC_JT 16
static void mlp2_dot_rows_blocked(int prec, uint32_t din, uint32_t h1n, uint32_t h2n, const float* x /* [ORC_MB][din] */,
                                  const float* c1, const float* w1, const float* w2, const float* b2, int act2,
                                  const float* w3, float bias3, float* xr /* [ORC_MB][din] */, float* h1buf /* [ORC_MB][h1n] */,
                                  float* h2buf /* [ORC_MB][h2n] */, float* out /* [ORC_MB] */) {
    for (uint32_t b = 0; b < ORC_MB; ++b)
        for (uint32_t k = 0; k < din; ++k) xr[(size_t)b * din + k] = op_round(x[(size_t)b * din + k], prec);
    /* (AVX2: _mm256_fmadd_ps is the correctly rounded fused multiply-add of fmaf, eight lanes at a time) */
    for (uint32_t j0 = 0; j0 < h1n; j0 += ORC_JT) {
        __m256 acc[ORC_MB][2];
        for (uint32_t b = 0; b < ORC_MB; ++b) {
            acc[b][0] = _mm256_loadu_ps(c1 + j0);
            acc[b][1] = _mm256_loadu_ps(c1 + j0 + 8);
        }
        for (uint32_t k = 0; k < din; ++k) {
            const float* wr = w1 + (size_t)k * h1n + j0;
            const __m256 w0 = _mm256_loadu_ps(wr), w1v = _mm256_loadu_ps(wr + 8);
            for (uint32_t b = 0; b < ORC_MB; ++b) {
                const __m256 xv = _mm256_broadcast_ss(xr + (size_t)b * din + k);
                acc[b][0] = _mm256_fmadd_ps(xv, w0, acc[b][0]);
                acc[b][1] = _mm256_fmadd_ps(xv, w1v, acc[b][1]);
            }
        }
        for (uint32_t b = 0; b < ORC_MB; ++b) {
            float tmp[ORC_JT];
            _mm256_storeu_ps(tmp, acc[b][0]);
            _mm256_storeu_ps(tmp + 8, acc[b][1]);
            for (uint32_t t = 0; t < ORC_JT; ++t) h1buf[(size_t)b * h1n + j0 + t] = op_round(tmp[t] > 0.0f ? tmp[t] : 0.0f, prec);
        }
    }
    for (uint32_t m0 = 0; m0 < h2n; m0 += ORC_JT) {
        __m256 acc[ORC_MB][2];
        for (uint32_t b = 0; b < ORC_MB; ++b) {
            acc[b][0] = _mm256_loadu_ps(b2 + m0);
            acc[b][1] = _mm256_loadu_ps(b2 + m0 + 8);
        }
        for (uint32_t j = 0; j < h1n; ++j) {
            const float* wr = w2 + (size_t)j * h2n + m0;
            const __m256 w0 = _mm256_loadu_ps(wr), w1v = _mm256_loadu_ps(wr + 8);
            for (uint32_t b = 0; b < ORC_MB; ++b) {
                const __m256 hv = _mm256_broadcast_ss(h1buf + (size_t)b * h1n + j);
                acc[b][0] = _mm256_fmadd_ps(hv, w0, acc[b][0]);
                acc[b][1] = _mm256_fmadd_ps(hv, w1v, acc[b][1]);
            }
        }
        for (uint32_t b = 0; b < ORC_MB; ++b) {
            float tmp[ORC_JT];
            _mm256_storeu_ps(tmp, acc[b][0]);
            _mm256_storeu_ps(tmp + 8, acc[b][1]);
            for (uint32_t t = 0; t < ORC_JT; ++t) h2buf[(size_t)b * h2n + m0 + t] = (act2 && !(tmp[t] > 0.0f)) ? 0.0f : tmp[t];
        }
    }
    const uint32_t half = h2n / 2;
    for (uint32_t b = 0; b < ORC_MB; ++b) {
        const float* hb = h2buf + (size_t)b * h2n;
        float p0 = bias3, p1 = 0.0f;
        for (uint32_t m = 0; m < half; ++m) p0 = fmaf(hb[m], w3[m], p0);
        for (uint32_t m = half; m < h2n; ++m) p1 = fmaf(hb[m], w3[m], p1);
        out[b] = sigmoidf_spec(p0 + p1);
    }
}

static float* round_copy(const float* w, size_t n, int prec) {
    float* o = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) o[i] = op_round(w[i], prec);
    return o;
}

/* DNN3 (cfg 3): input = [user_vec ‖ item_row]; the user half of layer 1 is request-constant:   */
/*   c1_j = chain(b1_j; P(u_k) * P(W1[k][j]), k asc over the user half)                         */
/* and the item half continues the same chain (k asc), so z1 is one 256-long k-ordered chain.   */
void orc_dnn3_forward(const orc_dnn3* m, int prec, const float* user_vec, const float* item_rows,
                      uint64_t n, float* out_scores, int threads) {
    const uint32_t du = m->d_user, di = m->d_item, h1 = m->h1, h2 = m->h2;
    float* w1 = round_copy(m->w1, (size_t)(du + di) * h1, prec);
    float* w2 = round_copy(m->w2, (size_t)h1 * h2, prec);
    float* c1 = (float*)malloc(h1 * sizeof(float));
    for (uint32_t j = 0; j < h1; ++j) c1[j] = m->b1[j];
    for (uint32_t k = 0; k < du; ++k) {
        const float uv = op_round(user_vec[k], prec);
        for (uint32_t j = 0; j < h1; ++j) c1[j] = fmaf(uv, w1[(size_t)k * h1 + j], c1[j]);
    }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const int blocked = threads >= 0 && h1 % ORC_JT == 0 && h2 % ORC_JT == 0;      /* threads < 0: the row-wise form (tests compare the two) */
#pragma omp parallel
    {
        float* hb1 = (float*)malloc((size_t)ORC_MB * h1 * sizeof(float));
        float* hb2 = (float*)malloc((size_t)ORC_MB * h2 * sizeof(float));
        float* xr = (float*)malloc((size_t)ORC_MB * di * sizeof(float));
        const int64_t nb = blocked ? (int64_t)(n / ORC_MB) : 0;
#pragma omp for schedule(static) nowait
        for (int64_t g = 0; g < nb; ++g)
            mlp2_dot_rows_blocked(prec, di, h1, h2, item_rows + (size_t)g * ORC_MB * di, c1, w1 + (size_t)du * h1, w2, m->b2, 1, m->w3,
                                  m->b3, xr, hb1, hb2, out_scores + g * ORC_MB);
#pragma omp for schedule(static)
        for (int64_t i = nb * ORC_MB; i < (int64_t)n; ++i)
            mlp2_dot_row(prec, di, h1, h2, item_rows + (size_t)i * di, c1, w1 + (size_t)du * h1, w2,
                         m->b2, 1, m->w3, m->b3, hb1, hb2, out_scores + i);
        free(xr); free(hb1); free(hb2);
    }
    free(c1); free(w2); free(w1);
}

/* FM + two-tower (cfg 4), SURVEY.md §8(d):                                                     */
/*   y_fm = b + Σ_f lin_f[id_f] + ½ Σ_k [ (Σ_f v_fk)² − Σ_f v_fk² ]   over 8 user + 8 item      */
/*   fields (user fields first), all fp32:                                                      */
/*     lin: sequential adds, fields in order;  s_k: sequential adds;  q_k = chain(0; v*v);      */
/*     t_k = fmaf(s_k, s_k, -q_k);  cross = balanced pairwise tree over k (xor 1,2,4,8);        */
/*     y_fm = lin + 0.5f*cross.                                                                 */
/*   user tower: u1 = P(relu(chain(ub1; P(u)*P(uw1)))), uo = chain(ub2; u1*P(uw2))  (no act)    */
/*   item tower: x = concat_f v_f (item fields), same shape via mlp2 (act2 = 0),                */
/*   score = σ( y_fm + <uo, io> )  with the dot split in two half-chains (see mlp2_dot_row).    */
/* user tower alone: the user embedding a vector model serves (online_vector_recall.go:97-109)       */
void orc_fm2t_user_embedding(const orc_fm2t* m, int prec, const float* user_vec, float* uo) {
    const uint32_t du = m->d_user, th = m->t_h1, to = m->t_out;
    float* uw1 = round_copy(m->uw1, (size_t)du * th, prec);
    float* uw2 = round_copy(m->uw2, (size_t)th * to, prec);
    float* u1 = (float*)malloc(th * sizeof(float));
    for (uint32_t j = 0; j < th; ++j) u1[j] = m->ub1[j];
    for (uint32_t k = 0; k < du; ++k) {
        const float uv = op_round(user_vec[k], prec);
        for (uint32_t j = 0; j < th; ++j) u1[j] = fmaf(uv, uw1[(size_t)k * th + j], u1[j]);
    }
    for (uint32_t j = 0; j < th; ++j) u1[j] = op_round(u1[j] > 0.0f ? u1[j] : 0.0f, prec);
    for (uint32_t o = 0; o < to; ++o) uo[o] = m->ub2[o];
    for (uint32_t j = 0; j < th; ++j)
        for (uint32_t o = 0; o < to; ++o) uo[o] = fmaf(u1[j], uw2[(size_t)j * to + o], uo[o]);
    free(u1); free(uw2); free(uw1);
}

void orc_fm2t_forward(const orc_fm2t* m, int prec, const float* const* field_emb,
                      const float* const* field_lin, const float* user_vec,
                      const int32_t* user_field_ids, const int32_t* item_field_ids, uint64_t n,
                      float* out_scores, int threads) {
    const uint32_t nuf = m->n_user_fields, nif = m->n_item_fields, K = m->k;
    const uint32_t th = m->t_h1, to = m->t_out, din = nif * K;
    /* user tower */
    float* iw1 = round_copy(m->iw1, (size_t)din * th, prec);
    float* iw2 = round_copy(m->iw2, (size_t)th * to, prec);
    float* uo = (float*)malloc(to * sizeof(float));
    orc_fm2t_user_embedding(m, prec, user_vec, uo);
    /* user prefix of the FM sums */
    float linU = m->fm_b;
    float* sU = (float*)calloc(K, sizeof(float));
    float* qU = (float*)calloc(K, sizeof(float));
    for (uint32_t f = 0; f < nuf; ++f) {
        const int32_t id = user_field_ids[f];
        linU = linU + field_lin[f][id];
        const float* v = field_emb[f] + (size_t)id * K;
        for (uint32_t k = 0; k < K; ++k) { sU[k] = sU[k] + v[k]; qU[k] = fmaf(v[k], v[k], qU[k]); }
    }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        float* x = (float*)malloc(din * sizeof(float));
        float* hb1 = (float*)malloc(th * sizeof(float));
        float* hb2 = (float*)malloc(to * sizeof(float));
        float* s = (float*)malloc(K * sizeof(float));
        float* q = (float*)malloc(K * sizeof(float));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t)n; ++i) {
            float lin = linU;
            for (uint32_t k = 0; k < K; ++k) { s[k] = sU[k]; q[k] = qU[k]; }
            for (uint32_t f = 0; f < nif; ++f) {
                const int32_t id = item_field_ids[(size_t)i * nif + f];
                lin = lin + field_lin[nuf + f][id];
                const float* v = field_emb[nuf + f] + (size_t)id * K;
                for (uint32_t k = 0; k < K; ++k) {
                    s[k] = s[k] + v[k];
                    q[k] = fmaf(v[k], v[k], q[k]);
                    x[f * K + k] = v[k];
                }
            }
            for (uint32_t k = 0; k < K; ++k) s[k] = fmaf(s[k], s[k], -q[k]);
            for (uint32_t off = 1; off < K; off <<= 1)           /* butterfly: all lanes equal */
                for (uint32_t k = 0; k < K; k += 2 * off) s[k] = s[k] + s[k + off];
            const float yfm = lin + 0.5f * s[0];
            mlp2_dot_row(prec, din, th, to, x, m->ib1, iw1, iw2, m->ib2, 0, uo, yfm, hb1, hb2,
                         out_scores + i);
        }
        free(q); free(s); free(hb2); free(hb1); free(x);
    }
    free(qU); free(sU); free(uo); free(iw2); free(iw1);
}

/* ------------------------------------------------------------------------------------------ */
/* sorts: sort/item_score.go:15-18 (Less = a.Score < b.Score), ItemScoreSort ascending (:36-41) */
/* ItemRankScoreSort = sort.Reverse → descending (item_rank_score.go:26-32).  Go's sort.Sort is */
/* unstable, so the reference leaves tie order undefined; this restatement fixes it to input    */
/* index ascending, and NaN last.                                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { double s; uint32_t i; } sort_ent;
static int g_sort_desc;
static int cmp_sort_ent(const void* a, const void* b) {
    const sort_ent* x = (const sort_ent*)a; const sort_ent* y = (const sort_ent*)b;
    const int xn = x->s != x->s, yn = y->s != y->s;
    if (xn || yn) { if (xn != yn) return xn - yn; return (x->i > y->i) - (x->i < y->i); }
    if (x->s < y->s) return g_sort_desc ? 1 : -1;
    if (x->s > y->s) return g_sort_desc ? -1 : 1;
    return (x->i > y->i) - (x->i < y->i);
}
void orc_sort_scores(const double* scores, uint32_t n, int desc, uint32_t* out_order) {
    sort_ent* e = (sort_ent*)malloc((size_t)n * sizeof(sort_ent));
    for (uint32_t i = 0; i < n; ++i) { e[i].s = scores[i]; e[i].i = i; }
    g_sort_desc = desc;
    qsort(e, n, sizeof(sort_ent), cmp_sort_ent);
    for (uint32_t i = 0; i < n; ++i) out_order[i] = e[i].i;
    free(e);
}

/* ------------------------------------------------------------------------------------------ */
/* DPP: sort/dpp_sort.go:372-551.                                                              */
/*   f_i = [e_i, 1] * (1/√2)                       (:428-430; e_i already L2-normalised)        */
/*   S   = F·Fᵀ                                    (:463-464)  — gonum Dgemm order is unknown   */
/*         here (un-vendored gonum v0.12.0): this restatement uses chain(0; f_ik*f_jk, k asc).  */
/*   L   = (r_i * S_ij) * r_j,  r = exp(alpha*rel) (:466-472; the two dense-diagonal Mul calls  */
/*         reduce to exactly these two roundings in that association).                          */
/*   greedy MAP with windows (:477-551), fp64, NaN-masked argmax (first max wins), ss summed    */
/*   sequentially over earlier picks with separate multiply and add.                            */
/* ------------------------------------------------------------------------------------------ */
void orc_l2_normalize_f64(double* v, uint32_t d) {
    /* floats.Norm(v,2) then floats.Scale(1/norm, v)  (dpp_sort.go:235-236) */
    double ss = 0.0;
    for (uint32_t k = 0; k < d; ++k) ss = fma(v[k], v[k], ss);
    const double inv = 1.0 / sqrt(ss);
    for (uint32_t k = 0; k < d; ++k) v[k] = inv * v[k];
}

void orc_dpp_kernel_matrix(const double* emb, uint32_t n, uint32_t d, const double* rel,
                           double alpha, double* L) {
    const double isq2 = 0.70710678118654757;      /* Go constant 1/math.Sqrt2, rounded to f64 */
    double* F = (double*)malloc((size_t)n * (d + 1) * sizeof(double));
    double* r = (double*)malloc((size_t)n * sizeof(double));
    for (uint32_t i = 0; i < n; ++i) {
        for (uint32_t k = 0; k < d; ++k) F[(size_t)i * (d + 1) + k] = isq2 * emb[(size_t)i * d + k];
        F[(size_t)i * (d + 1) + d] = isq2 * 1.0;
        r[i] = exp(alpha * rel[i]);
    }
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < (int32_t)n; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            double s = 0.0;
            const double* a = F + (size_t)i * (d + 1);
            const double* b = F + (size_t)j * (d + 1);
            for (uint32_t k = 0; k <= d; ++k) s = fma(a[k], b[k], s);
            L[(size_t)i * n + j] = (r[i] * s) * r[j];
        }
    free(r); free(F);
}

/* The same from finished feature rows F [n][d1] (KernelMatrix's featureMat, dpp_sort.go:407-461): covers the     */
/* hook-embedding and EnsurePositiveSim = false variants, whose rows are built by oracle.py:dpp_features.         */
void orc_dpp_kernel_matrix_f(const double* F, uint32_t n, uint32_t d1, const double* rel, double alpha, double* L) {
    double* r = (double*)malloc((size_t)n * sizeof(double));
    for (uint32_t i = 0; i < n; ++i) r[i] = exp(alpha * rel[i]);
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < (int32_t)n; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            double s = 0.0;
            const double* a = F + (size_t)i * d1;
            const double* b = F + (size_t)j * d1;
            for (uint32_t k = 0; k < d1; ++k) s = fma(a[k], b[k], s);
            L[(size_t)i * n + j] = (r[i] * s) * r[j];
        }
    free(r);
}

static int idx_in(const uint32_t* a, uint32_t n, uint32_t e) {
    for (uint32_t i = 0; i < n; ++i) if (a[i] == e) return 1;
    return 0;
}
static uint32_t max_idx_nan_skip(const double* v, uint32_t n) {   /* gonum floats.MaxIdx */
    double mx = NAN; uint32_t ind = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (v[i] != v[i]) continue;
        if (v[i] > mx || mx != mx) { mx = v[i]; ind = i; }
    }
    return ind;
}

static uint32_t dpp_once(const double* L, uint32_t N, uint32_t topn, const uint32_t* existed,
                         uint32_t n_existed, uint32_t* Y) {
    const double epsilon = 1e-10;
    if (topn > N) topn = N;
    if (topn == 0) return 0;
    double* d2 = (double*)malloc((size_t)N * sizeof(double));
    double* c = (double*)calloc((size_t)topn * N, sizeof(double));
    double* e = (double*)malloc((size_t)N * sizeof(double));
    uint32_t ny = 0;
    for (uint32_t i = 0; i < N; ++i) d2[i] = idx_in(existed, n_existed, i) ? NAN : L[(size_t)i * N + i];
    uint32_t j = max_idx_nan_skip(d2, N);
    Y[ny++] = j;
    while (ny < topn) {
        double dj = d2[j];
        if (dj < epsilon) break;
        dj = sqrt(dj);
        const uint32_t k = ny - 1;
        const double inv = 1.0 / dj;
        for (uint32_t n = 0; n < N; ++n) {
            double lj = L[(size_t)j * N + n];
            if (k > 0) {
                double ss = 0.0;
                for (uint32_t i = 0; i < k; ++i) {
                    volatile double p = c[(size_t)i * N + j] * c[(size_t)i * N + n];  /* no FMA */
                    ss = ss + p;
                }
                lj = lj - ss;
            }
            e[n] = inv * lj;
        }
        for (uint32_t n = 0; n < N; ++n) {
            c[(size_t)k * N + n] = e[n];
            volatile double e2 = e[n] * e[n];
            d2[n] = d2[n] - e2;
        }
        d2[j] = NAN;
        j = max_idx_nan_skip(d2, N);
        Y[ny++] = j;
    }
    if (ny < topn)
        for (uint32_t i = 0; i < N && ny < topn; ++i)
            if (!idx_in(existed, n_existed, i) && !idx_in(Y, ny, i)) Y[ny++] = i;
    free(e); free(c); free(d2);
    return ny;
}

uint32_t orc_dpp_with_window(const double* L, uint32_t n, uint32_t topn, uint32_t window,
                             uint32_t* out_idx) {
    /* DPPWithWindow, dpp_sort.go:477-491 */
    if (topn <= window) return dpp_once(L, n, topn, NULL, 0, out_idx);
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < topn / window; ++i) cnt += dpp_once(L, n, window, out_idx, cnt, out_idx + cnt);
    if (topn % window) cnt += dpp_once(L, n, topn % window, out_idx, cnt, out_idx + cnt);
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* SSD: sort/ssd_sort.go:346-486 (SSDWithSlidingWindow, arXiv 2107.05204), fp64.               */
/* gonum v0.12.0 (floats.Dot / floats.Norm / mat.ScaleVec / stat.PopMeanVariance) is not        */
/* vendored in the reference tree and its amd64 kernels are assembly, so the summation orders   */
/* below are this restatement's specification (parity unpinned at that boundary):               */
/*   dot(a,b)  = chain(0; a_k*b_k, k asc)  (fma)          norm(v) = sqrt(chain(0; v_k*v_k))      */
/*   e -= p*f  = e_k - (p*f_k)  (ScaleVec then floats.Sub: two roundings); += likewise           */
/*   quality   = r + (volume*l2)            argmax = floats.MaxIdx (first max, NaN skipped)      */
/* ------------------------------------------------------------------------------------------ */
/* quality-score normalisation (:360-388).  mode 0: copy; 1: z-score (stat.PopMeanVariance,     */
/* two-pass with compensation; stat.StdScore); 2: min-max into [eps,1] with max = rel[0],       */
/* min = rel[n-1] (items arrive sorted by score, descending).  Returns 0 when the reference      */
/* bails out ("all item score are zeros": mean==0 || variance==0, or span==0).                   */
int orc_ssd_quality(const double* rel, uint32_t n, int mode, double* out) {
    if (mode == 1) {
        double sum = 0.0;
        for (uint32_t i = 0; i < n; ++i) sum = sum + rel[i];
        const double mean = sum / (double)n;
        double ss = 0.0, comp = 0.0;
        for (uint32_t i = 0; i < n; ++i) {
            const double d = rel[i] - mean;
            volatile double dd = d * d;           /* no FMA */
            ss = ss + dd;
            comp = comp + d;
        }
        volatile double cc = comp * comp;
        const double variance = (ss - cc / (double)n) / (double)n;
        if (mean == 0.0 || variance == 0.0) return 0;
        const double sd = sqrt(variance);
        for (uint32_t i = 0; i < n; ++i) out[i] = (rel[i] - mean) / sd;
        return 1;
    }
    if (mode == 2) {
        const double mx = rel[0], mn = rel[n - 1], span = mx - mn;
        if (span == 0.0) return 0;
        const double eps = 1e-6;
        for (uint32_t i = 0; i < n; ++i) {
            volatile double a = ((rel[i] - mn) / span) * (1 - eps);
            out[i] = a + eps;
        }
        return 1;
    }
    for (uint32_t i = 0; i < n; ++i) out[i] = rel[i];
    return 1;
}

static double ssd_norm(const double* v, uint32_t d) {
    double ss = 0.0;
    for (uint32_t k = 0; k < d; ++k) ss = fma(v[k], v[k], ss);
    return sqrt(ss);
}

/* emb: [n][d] fp64, modified in place (the reference mutates Item.Embedding).  Returns the     */
/* number of indices written (min(n, topn)).                                                    */
uint32_t orc_ssd_window(double* emb, uint32_t n, uint32_t d, const double* rel, double gamma,
                        uint32_t topn, uint32_t window, int use_ssd_star, uint32_t* out_idx) {
    if (n == 0 || topn == 0) return 0;
    if (window <= 1) window = 5;                                    /* :357-360 */
    const uint32_t T = n < topn ? n : topn;
    uint8_t* selected = (uint8_t*)calloc(n, 1);
    double* proj = (double*)calloc((size_t)window * n, sizeof(double));   /* ring of projection vectors */
    double* q = (double*)malloc((size_t)n * sizeof(double));
    uint32_t t = 1;
    uint32_t idx = max_idx_nan_skip(rel, n);
    selected[idx] = 1;
    out_idx[0] = idx;
    double volume = gamma;
    if (!use_ssd_star) {
        const double l2 = ssd_norm(emb + (size_t)idx * d, d);
        if (!(isnan(l2) || isinf(l2))) volume *= l2;
    }
    while (t < T) {
        const uint32_t slot = t % window;
        if (t > window) {                         /* restore the projection of the item leaving the window */
            const uint32_t i = out_idx[t - 1 - window];
            const double* ei = emb + (size_t)i * d;
            const double* pold = proj + (size_t)slot * n;
            for (uint32_t j = 0; j < n; ++j) {
                if (selected[j]) continue;
                double* ej = emb + (size_t)j * d;
                for (uint32_t k = 0; k < d; ++k) {
                    volatile double s = pold[j] * ei[k];
                    ej[k] = ej[k] + s;
                }
            }
        }
        const double* es = emb + (size_t)idx * d;
        double den = 0.0;
        for (uint32_t k = 0; k < d; ++k) den = fma(es[k], es[k], den);
        double* pnew = proj + (size_t)slot * n;
        for (uint32_t j = 0; j < n; ++j) {
            pnew[j] = 0.0;
            if (selected[j]) continue;
            double* ej = emb + (size_t)j * d;
            double acc = 0.0;
            for (uint32_t k = 0; k < d; ++k) acc = fma(ej[k], es[k], acc);
            double p = acc / den;
            if (isnan(p) || isinf(p)) p = 1.0;
            pnew[j] = p;
            for (uint32_t k = 0; k < d; ++k) {
                volatile double s = p * es[k];
                ej[k] = ej[k] - s;
            }
        }
        ++t;
        for (uint32_t j = 0; j < n; ++j) {
            if (selected[j]) { q[j] = -DBL_MAX; continue; }
            const double l2 = ssd_norm(emb + (size_t)j * d, d);
            volatile double v = volume * ((isnan(l2) || isinf(l2)) ? 0.5 : l2);
            q[j] = rel[j] + v;
        }
        idx = max_idx_nan_skip(q, n);
        selected[idx] = 1;
        out_idx[t - 1] = idx;
        if (!use_ssd_star) {
            const double l2 = ssd_norm(emb + (size_t)idx * d, d);
            if (!(isnan(l2) || isinf(l2))) volume *= l2;
        }
    }
    free(q); free(proj); free(selected);
    return T;
}
