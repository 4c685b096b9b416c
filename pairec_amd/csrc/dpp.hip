// dpp.hip — DPP diversity re-rank in fp64 on the device.
//
// Replaces DPPSort.KernelMatrix + DPPWithWindow + DPP (sort/dpp_sort.go:372-551, built on gonum
// v0.12.0 mat.Dense.Mul / floats.*).  The reference materialises
//     F = [e_i/‖e_i‖ , 1] / √2,   S = F·Fᵀ,   L = diag(r)·S·diag(r),  r_i = exp(α·score_i)
// with two dense N×N×N multiplies by a diagonal matrix (dpp_sort.go:463-472); those reduce to
//     L_ij = (r_i · S_ij) · r_j
// which is what is computed here (2 roundings, same association), then runs greedy MAP inference
// in windows with an incremental Cholesky update and a NaN-masked argmax.
//
// Summation orders: S_ij = chain_{k asc} fma(F_ik, F_jk, ·) — gonum's Dgemm order
// is unknowable here (module not vendored), parity with the reference is unpinned at that boundary;
// the greedy update uses separate multiply and add, sequential over earlier picks, exactly as
// gonum's Dgemm-by-axpy does on amd64, and is bit-identical to oracle/oracle.c given the same L.
#include "common.hpp"

#include <cmath>

namespace pg {

// one thread per candidate: gather the fp32 embedding, widen, optionally L2-normalise
// (floats.Norm / floats.Scale(1/norm), dpp_sort.go:235-236), build F row and r_i.
__global__ void dpp_prepare_kernel(const float* __restrict__ tab, uint32_t tab_rows, uint32_t d,
                                   const uint32_t* __restrict__ cand, const double* __restrict__ rel,
                                   uint32_t n, double alpha, int normalize, double* __restrict__ F,
                                   double* __restrict__ r) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t row = cand[i];
    row = row < tab_rows ? row : tab_rows - 1;
    const float* x = tab + (size_t)row * d;
    double inv = 1.0;
    if (normalize) {
        double ss = 0.0;
        for (uint32_t k = 0; k < d; ++k) {
            const double v = (double)x[k];
            ss = fma(v, v, ss);
        }
        inv = 1.0 / sqrt(ss);
    }
    const double isq2 = 0.70710678118654757;      // Go constant 1/math.Sqrt2
    double* f = F + (size_t)i * (d + 1);
    for (uint32_t k = 0; k < d; ++k) {
        double v = (double)x[k];
        if (normalize) v = inv * v;
        f[k] = isq2 * v;
    }
    f[d] = isq2 * 1.0;
    r[i] = exp(alpha * rel[i]);
}

__global__ void dpp_kernel_matrix_kernel(const double* __restrict__ F, const double* __restrict__ r,
                                         uint32_t n, uint32_t d1, double* __restrict__ L) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = blockIdx.y;
    if (j >= n) return;
    const double* a = F + (size_t)i * d1;
    const double* b = F + (size_t)j * d1;
    double s = 0.0;
    for (uint32_t k = 0; k < d1; ++k) s = fma(a[k], b[k], s);
    L[(size_t)i * n + j] = __dmul_rn(__dmul_rn(r[i], s), r[j]);
}

// floats.MaxIdx: first maximum, NaN skipped; all-NaN → index 0.  Block-wide, result in *s_idx.
__device__ __forceinline__ void block_argmax(const double* __restrict__ v, uint32_t n, double* s_val,
                                             uint32_t* s_idx, uint32_t* out_idx) {
    const uint32_t tid = threadIdx.x;
    double best = 0.0;
    uint32_t bi = 0xFFFFFFFFu;                     // "none yet"
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const double x = v[i];
        if (x != x) continue;
        if (bi == 0xFFFFFFFFu || x > best) { best = x; bi = i; }   // ascending i per thread → first max
    }
    s_val[tid] = best;
    s_idx[tid] = bi;
    __syncthreads();
    for (uint32_t s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s) {
            const uint32_t oi = s_idx[tid + s];
            const double ov = s_val[tid + s];
            const uint32_t mi = s_idx[tid];
            const double mv = s_val[tid];
            if (oi != 0xFFFFFFFFu && (mi == 0xFFFFFFFFu || ov > mv || (ov == mv && oi < mi))) {
                s_val[tid] = ov;
                s_idx[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0) *out_idx = (s_idx[0] == 0xFFFFFFFFu) ? 0u : s_idx[0];
    __syncthreads();
}

// DPPWithWindow + DPP (dpp_sort.go:477-551), one workgroup.
//   d2: [n], c: [window][n] scratch in global memory (L2-resident), Y: output indices.
__global__ __launch_bounds__(1024) void dpp_greedy_kernel(const double* __restrict__ L, uint32_t N,
                                                          uint32_t topn_total, uint32_t window,
                                                          double* __restrict__ d2, double* __restrict__ c,
                                                          uint32_t* __restrict__ out, uint32_t* __restrict__ out_count) {
    __shared__ double s_val[1024];
    __shared__ uint32_t s_idx[1024];
    __shared__ uint32_t s_j;
    const uint32_t tid = threadIdx.x;
    const double epsilon = 1e-10;
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    uint32_t done = 0;                              // len(result) so far
    // window schedule: topN <= window → one call; else topN/window calls + remainder
    uint32_t n_calls, rem;
    if (topn_total <= window) { n_calls = 1; rem = 0; }
    else { n_calls = topn_total / window; rem = topn_total % window; }
    for (uint32_t call = 0; call < n_calls + (rem ? 1u : 0u); ++call) {
        uint32_t topn = (topn_total <= window) ? topn_total : (call < n_calls ? window : rem);
        if (topn > N) topn = N;
        if (topn == 0) continue;
        uint32_t* Y = out + done;
        const uint32_t existed = done;
        // d2[i] = L_ii, NaN for already selected
        for (uint32_t i = tid; i < N; i += blockDim.x) {
            bool ex = false;
            for (uint32_t e = 0; e < existed; ++e) ex |= (out[e] == i);
            d2[i] = ex ? nan : L[(size_t)i * N + i];
        }
        __syncthreads();
        block_argmax(d2, N, s_val, s_idx, &s_j);
        uint32_t j = s_j;
        uint32_t ny = 0;
        if (tid == 0) Y[0] = j;
        ny = 1;
        bool broke = false;
        while (ny < topn) {
            double dj = d2[j];
            __syncthreads();                            // everyone has read d2[j] before it is updated
            if (dj < epsilon) { broke = true; break; }
            dj = sqrt(dj);
            const uint32_t k = ny - 1;
            const double inv = 1.0 / dj;
            for (uint32_t n = tid; n < N; n += blockDim.x) {
                double lj = L[(size_t)j * N + n];
                if (k > 0) {
                    double ss = 0.0;
                    for (uint32_t i = 0; i < k; ++i)
                        ss = __dadd_rn(ss, __dmul_rn(c[(size_t)i * N + j], c[(size_t)i * N + n]));
                    lj = __dsub_rn(lj, ss);
                }
                const double e = __dmul_rn(inv, lj);
                c[(size_t)k * N + n] = e;
                d2[n] = __dsub_rn(d2[n], __dmul_rn(e, e));
            }
            __syncthreads();
            if (tid == 0) d2[j] = nan;
            __syncthreads();
            block_argmax(d2, N, s_val, s_idx, &s_j);
            j = s_j;
            if (tid == 0) Y[ny] = j;
            ++ny;
            __syncthreads();
        }
        __syncthreads();
        if (broke && ny < topn) {
            if (tid == 0) {
                for (uint32_t i = 0; i < N && ny < topn; ++i) {
                    bool used = false;
                    for (uint32_t e = 0; e < existed + ny; ++e) used |= (out[e] == i);
                    if (!used) Y[ny++] = i;
                }
                s_j = ny;
            }
            __syncthreads();
            ny = s_j;
        }
        done += ny;
        __syncthreads();
    }
    if (tid == 0) *out_count = done;
}

}  // namespace pg

extern "C" {

int pg_dpp(pg_ctx* ctx, const pg_table* t, const uint32_t* cand_rows, const double* rel, uint32_t n,
           double alpha, uint32_t topn, uint32_t window, int normalize_emb, uint32_t* out_idx,
           uint32_t* out_count) {
    PG_REQUIRE(ctx && t && out_count, "pg_dpp: NULL argument");
    *out_count = 0;
    if (n == 0 || topn == 0) return PG_OK;
    PG_REQUIRE(cand_rows && rel && out_idx, "pg_dpp: NULL argument");
    if (window == 0) window = 10;                        // NewDPPSort default (dpp_sort.go:89-91)
    if (n > 8192) {
        pg::set_error("pg_dpp: %u candidates unsupported (<= 8192; the reference caps N with CandidateCount)", n);
        return PG_ERR_UNSUPPORTED;
    }
    for (uint32_t i = 0; i < n; ++i)
        PG_REQUIRE(cand_rows[i] < t->rows, "pg_dpp: candidate row %u outside table", cand_rows[i]);
    std::lock_guard<std::mutex> g(ctx->mu);
    const uint32_t d1 = t->dim + 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t bF = al((size_t)n * d1 * 8), bR = al((size_t)n * 8), bL = al((size_t)n * n * 8);
    const size_t bD2 = al((size_t)n * 8), bC = al((size_t)std::min(window, n) * n * 8);
    const size_t bCand = al((size_t)n * 4), bRel = al((size_t)n * 8), bOut = al((size_t)(topn + 1) * 4 + 16);
    void* buf;
    int rc;
    if ((rc = pg::scratch_reserve(ctx, 7, bF + bR + bL + bD2 + bC + bCand + bRel + bOut, &buf))) return rc;
    char* p = (char*)buf;
    double* F = (double*)p; p += bF;
    double* R = (double*)p; p += bR;
    double* L = (double*)p; p += bL;
    double* D2 = (double*)p; p += bD2;
    double* Cm = (double*)p; p += bC;
    uint32_t* d_cand = (uint32_t*)p; p += bCand;
    double* d_rel = (double*)p; p += bRel;
    uint32_t* d_out = (uint32_t*)p;
    uint32_t* d_cnt = d_out + topn;
    PG_HIP(hipMemcpyAsync(d_cand, cand_rows, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_rel, rel, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    pg::dpp_prepare_kernel<<<(n + 63) / 64, 64, 0, ctx->stream>>>(t->d, (uint32_t)t->rows, t->dim, d_cand, d_rel, n,
                                                                alpha, normalize_emb, F, R);
    pg::dpp_kernel_matrix_kernel<<<dim3((n + 255) / 256, n), 256, 0, ctx->stream>>>(F, R, n, d1, L);
    pg::dpp_greedy_kernel<<<1, 1024, 0, ctx->stream>>>(L, n, topn, window, D2, Cm, d_out, d_cnt);
    PG_HIP(hipGetLastError());
    PG_HIP(hipMemcpyAsync(ctx->h_status + 330, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    const uint32_t cnt = ctx->h_status[330];
    PG_HIP(hipMemcpyAsync(out_idx, d_out, (size_t)cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    *out_count = cnt;
    return PG_OK;
}

}  // extern "C"
