// recall_r2.hip — the refinement stage between the int8 screen and the exact re-scoring, for tables whose rows CROWD.
//
// The int8 screen (recall.hip) bounds a score to eps_q ~ R ||q|| + (N + R) ||q - q^||: ~0.013 for unit rows and queries.  On
// trained embeddings that is not small: rows cluster, a query's top-K sits inside one cluster whose members score within a few
// thousandths of each other, and every member within eps_q of the K-th score is a suspect — 9 to 20 per answer on the
// clustered benchmark table (pg_table_fill_mixture; uniform and i.i.d. Gaussian rows: 1.8) — each a 512-B gather in
// rescore_kernel: 2.4 ms per 256-query pass at 100 M rows, as long as the screen itself.  Here every suspect is looked at once
// more with two more digits: the row's int8 shadow X8 plus an int8 RESIDUAL shadow Xr (x = s8 X8 + s8r Xr + e2, s8r = s8 / 254,
// ||e2|| <= R2 measured: ~R / 250), against the query in sixteen bits (q = sq16 Q16 + eq).  With U = sum X8 Q16, V = sum Xr Q16
// (exact integers):
//     s = sq16 (s8 U + s8r V) + sum x^ eq + sum e2 q,      |sum x^ eq| <= (N + R2) ||eq||,   |sum e2 q| <= R2 ||q||
// so a row can reach thr only if  sq16 (s8 U + s8r V) >= thr - eps2_q,  eps2_q = (N + R2) ||eq|| + R2 ||q|| + 1e-5 N ||q|| (the
// last term: the rounding of the specification's fp32 fmaf chain, as in eps_q) — two hundred times tighter than eps_q.  What
// passes is re-scored exactly as ever; the answers do not change, only the number of 512-B gathers (20 -> ~1.1 per answer).
// Cost: two 128-B lines per suspect instead of four, through rescreen8_kernel's pipelined gather (recall_i4m.hip), and 12.8 GB
// per 100 M rows for the residual shadow, built the first time a table shows more than Knobs::r2_min_factor suspects per answer.
// Reference path: the same VectorRecall.GetCandidateItems → FaissModel.Run top-K as recall.hip
// (service/recall/vector_recall.go:32-123; the tables behind service/recall/hologres_vector_recall.go:23 are trained embeddings).
#include "common.hpp"

namespace pg {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// the residual shadow and the largest second residual.  A thread converts 8 consecutive values; 16 neighbouring lanes share a row.
// X8 is recomputed exactly as table_quant8_kernel computed it (same instruction sequence), so Xr belongs to the stored shadow.
__global__ __launch_bounds__(256) void table_quant8r_kernel(const float* __restrict__ tab, uint64_t rows, float s, float sr,
                                                            int8_t* __restrict__ out8r, float* __restrict__ out_resid2) {
    constexpr int G = 16;
    __shared__ float smax[4];
    const uint64_t n8 = rows * (uint64_t)G;
    const float inv = 1.0f / s, invr = 1.0f / sr;
    float mx = 0.0f;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ((n8 + 63) & ~63ull); g += (uint64_t)gridDim.x * blockDim.x) {
        float rs = 0.0f;
        if (g < n8) {
            const float4 a = reinterpret_cast<const float4*>(tab)[2 * g];
            const float4 b = reinterpret_cast<const float4*>(tab)[2 * g + 1];
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t w[2] = {0, 0};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int X = __float2int_rn(v[i] * inv);
                X = X > 127 ? 127 : (X < -127 ? -127 : X);
                const float r = __fmaf_rn(-s, (float)X, v[i]);
                int Xr = __float2int_rn(r * invr);
                Xr = Xr > 127 ? 127 : (Xr < -127 ? -127 : Xr);
                const float e2 = __fmaf_rn(-sr, (float)Xr, r);
                rs = __fmaf_rn(e2, e2, rs);
                w[i >> 2] |= (uint32_t)(Xr & 0xff) << (8 * (i & 3));
            }
            reinterpret_cast<uint2*>(out8r)[g] = make_uint2(w[0], w[1]);
        }
#pragma unroll
        for (int off = 1; off < G; off <<= 1) rs += __shfl_xor(rs, off, 64);
        mx = fmaxf(mx, rs);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) smax[w] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<uint32_t*>(out_resid2), __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))));
}

// per call: the queries in sixteen bits as two int8 planes (Q16 = 256 Qh + Ql, Qh = floor((Q16 + 128) / 256), Ql in [-128, 127])
// in plain order, and per query {sq16, ||q||-side constants}: q16 = [256][32] Qh | [256][32] Ql | [256][4] {sq16, eps2 (double), -}
constexpr uint32_t kQlAt = kMaxQueries * 32, kQ16cAt = 2 * kMaxQueries * 32;
__global__ __launch_bounds__(256) void rescreen16_prep_kernel(const float* __restrict__ qpad, float max_norm, float resid2,
                                                              uint32_t* __restrict__ q16) {
    const uint32_t qi = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const float* q = qpad + (size_t)qi * 128 + part * 32;
    float mx = 0.0f;
    int bad = 0;
    for (int k = 0; k < 32; ++k) {
        const float v = fabsf(q[k]);
        if (!(v <= 3.0e38f)) bad = 1;
        mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    bad |= __shfl_xor(bad, 1, 64);
    bad |= __shfl_xor(bad, 2, 64);
    const float sc = fmaxf(mx / 32639.0f, 1e-30f);
    double ss = 0.0, dd = 0.0;
    uint32_t* const oh = q16 + (size_t)qi * 32 + part * 8;
    uint32_t* const ol = q16 + kQlAt + (size_t)qi * 32 + part * 8;
    for (int e = 0; e < 8; ++e) {
        uint32_t wh = 0, wl = 0;
        for (int b = 0; b < 4; ++b) {
            const float f = q[4 * e + b];
            int Q = __float2int_rn(f / sc);
            Q = Q > 32639 ? 32639 : (Q < -32639 ? -32639 : Q);
            if (bad) Q = 0;
            const int Qh = (Q + 128) >> 8, Ql = Q - 256 * Qh;
            const double v = (double)f, d = v - (double)sc * (double)Q;
            ss += v * v;
            dd += d * d;
            wh |= (uint32_t)(Qh & 0xff) << (8 * b);
            wl |= (uint32_t)(Ql & 0xff) << (8 * b);
        }
        oh[e] = wh;
        ol[e] = wl;
    }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    dd += __shfl_xor(dd, 1, 64);
    dd += __shfl_xor(dd, 2, 64);
    if (part == 0) {
        const double nrm = sqrt(ss), de = sqrt(dd), N = (double)max_norm, R2 = (double)resid2;
        // (+ 1e-9 N ||q||: the double evaluation of sq16 (s8 U + s8r V) in the kernel)
        const double eps2 = ((N + R2) * de + R2 * nrm) * 1.0001 + (1e-5 + 1e-9) * N * nrm + 1e-30;
        uint32_t* const c = q16 + kQ16cAt + (size_t)qi * 4;
        c[0] = __float_as_uint(sc);
        const double e = (bad || !(eps2 == eps2) || eps2 > 1e300) ? __builtin_inf() : eps2;       // (inf: every suspect passes)
        memcpy(c + 1, &e, 8);
        c[3] = 0;
    }
}

constexpr int kR2Stage = 448;
__device__ __forceinline__ void r2_gather(i32x4 (&v)[8], const int8_t* __restrict__ d, uint32_t row, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r_i = (uint32_t)__shfl((int)row, (lane & 56) + i, 64);
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(d + (size_t)r_i * 128) + (lane & 7));
    }
}
// the eight rows' eight partial sums of a lane group reduced across the group: lane 8 g + i ends with row i's total (recall_i4m.hip)
__device__ __forceinline__ int r2_transpose_sum(const int (&p)[8], int lane) {
    const bool b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    int s4[4], s2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) s4[i] = (b2 ? p[i + 4] : p[i]) + __shfl_xor(b2 ? p[i] : p[i + 4], 4, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) s2[i] = (b1 ? s4[i + 2] : s4[i]) + __shfl_xor(b1 ? s4[i] : s4[i + 2], 2, 64);
    return (b0 ? s2[1] : s2[0]) + __shfl_xor(b0 ? s2[0] : s2[1], 1, 64);
}
__device__ __forceinline__ int r2_dot16(const i32x4& x, const i32x4& qh, const i32x4& ql) {       // sum X (256 Qh + Ql) over 16 elements
    int h = __builtin_amdgcn_sdot4(x.x, qh.x, 0, false), l = __builtin_amdgcn_sdot4(x.x, ql.x, 0, false);
    h = __builtin_amdgcn_sdot4(x.y, qh.y, h, false);
    l = __builtin_amdgcn_sdot4(x.y, ql.y, l, false);
    h = __builtin_amdgcn_sdot4(x.z, qh.z, h, false);
    l = __builtin_amdgcn_sdot4(x.z, ql.z, l, false);
    h = __builtin_amdgcn_sdot4(x.w, qh.w, h, false);
    l = __builtin_amdgcn_sdot4(x.w, ql.w, l, false);
    return 256 * h + l;                                  // |.| <= 16 x 127 x 32639 < 2^27; a row's total < 2^30
}

// grid (blocks, nq), 256 threads: the suspects of every query (susp, [nq][scap]) that the two-digit bound still lets reach thr,
// compacted into susp2 ([nq][cap2]).  The gathers of the next chunk of 64 suspects are in flight under the arithmetic of this one.
__global__ __launch_bounds__(256) void rescreen16_kernel(const int8_t* __restrict__ d8, const int8_t* __restrict__ d8r, float s8, float s8r,
                                                         const uint32_t* __restrict__ q16, const float* __restrict__ thr,
                                                         const uint32_t* __restrict__ susp, const uint32_t* __restrict__ susp_cnt, uint32_t scap,
                                                         uint32_t n_rows, uint32_t* __restrict__ susp2, uint32_t* __restrict__ susp2_cnt,
                                                         uint32_t cap2, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t stage[4][kR2Stage];
    const uint32_t q = blockIdx.y;
    const uint32_t n_raw = susp_cnt[q];
    const uint32_t n = n_raw < scap ? n_raw : scap;
    if (n_raw > scap && blockIdx.x == 0 && threadIdx.x == 0) *overflow = 1u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const i32x4 qh = *reinterpret_cast<const i32x4*>(q16 + (size_t)q * 32 + (lane & 7) * 4);
    const i32x4 ql = *reinterpret_cast<const i32x4*>(q16 + kQlAt + (size_t)q * 32 + (lane & 7) * 4);
    const uint32_t* const c = q16 + kQ16cAt + (size_t)q * 4;
    const double sq16 = (double)__uint_as_float(c[0]);
    double eps2;
    memcpy(&eps2, c + 1, 8);
    const float t = thr[q];
    // M >= thr - eps2; an open or odd threshold, or constants that are not finite: everything passes
    const double cut = (t == t && t > -__builtin_inff() && eps2 < 1e300) ? (double)t - eps2 : -__builtin_inf();
    const double a8 = sq16 * (double)s8, a8r = sq16 * (double)s8r;
    const uint32_t nchunks = (n + 63) / 64, cstep = gridDim.x * 4;
    uint32_t ch = blockIdx.x * 4 + (uint32_t)w;
    if (ch >= nchunks) return;
    uint32_t cnt = 0;
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&susp2_cnt[q], cnt);
        base = __builtin_amdgcn_readfirstlane(base);
        for (uint32_t i = lane; i < cnt; i += 64) {
            const uint32_t pos = base + i;
            if (pos < cap2) susp2[(uint64_t)q * cap2 + pos] = stage[w][i];
            else *overflow = 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cnt = 0;
    };
    auto rows_of = [&](uint32_t c_) -> uint32_t {
        const uint32_t e = c_ * 64 + lane;
        const uint32_t r = (c_ < nchunks && e < n) ? susp[(uint64_t)q * scap + e] : 0u;
        return r < n_rows ? r : 0u;
    };
    auto emit = [&](uint32_t c_, uint32_t row, const i32x4 (&v8)[8], const i32x4 (&vr)[8]) {
        int pu[8], pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pu[i] = r2_dot16(v8[i], qh, ql);
            pv[i] = r2_dot16(vr[i], qh, ql);
        }
        const int U = r2_transpose_sum(pu, lane), V = r2_transpose_sum(pv, lane);
        const double M = a8 * (double)U + a8r * (double)V;
        const bool keep = c_ * 64 + lane < n && !(M < cut);
        const uint64_t bm = __builtin_amdgcn_ballot_w64(keep);
        if (bm) {
            const uint32_t k = (uint32_t)__popcll(bm);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (keep) stage[w][cnt + before] = row;
            cnt += k;
            if (cnt > (uint32_t)(kR2Stage - 64)) flush();
        }
    };
    i32x4 a0[8], a1[8], b0[8], b1[8];
    uint32_t rowA = rows_of(ch), rowB = rows_of(ch + cstep);
    r2_gather(a0, d8, rowA, lane);
    r2_gather(a1, d8r, rowA, lane);
    for (;;) {
        const bool hasB = ch + cstep < nchunks;
        if (hasB) {
            r2_gather(b0, d8, rowB, lane);
            r2_gather(b1, d8r, rowB, lane);
        }
        const uint32_t rowC = rows_of(ch + 2 * cstep);
        emit(ch, rowA, a0, a1);
        if (!hasB) break;
        ch += cstep;
        const bool hasC = ch + cstep < nchunks;
        if (hasC) {
            r2_gather(a0, d8, rowC, lane);
            r2_gather(a1, d8r, rowC, lane);
        }
        const uint32_t rowD = rows_of(ch + 2 * cstep);
        emit(ch, rowB, b0, b1);
        if (!hasC) break;
        ch += cstep;
        rowA = rowC;
        rowB = rowD;
    }
    if (cnt) flush();
}

std::mutex g_r2_build_mu;

}  // namespace

// the residual shadow of an int8-screened dim-128 table (lazily: the first batch after the table showed crowded rows)
int ensure_table_r2(pg_ctx* ctx, const pg_table* tc) {
    pg_table* t = const_cast<pg_table*>(tc);
    if (t->r2_ok || t->r2_failed) return PG_OK;
    std::lock_guard<std::mutex> g(g_r2_build_mu);
    if (t->r2_ok || t->r2_failed) return PG_OK;
    if (t->dim != 128 || !t->stats_valid || !t->all_finite || !t->shadow_is_i8 || !t->d8) { t->r2_failed = true; return PG_OK; }
    void* p;
    int rc;
    if ((rc = scratch_reserve(ctx, 4, 4096, &p))) return rc;
    float* d_st = (float*)p + 340;
    if (!t->d8r) {
        if (hipMalloc((void**)&t->d8r, (t->rows + 64) * (size_t)128) != hipSuccess) {
            (void)hipGetLastError();
            t->d8r = nullptr;
            t->r2_failed = true;                       // no memory: the suspects go to the exact re-scoring as before
            return PG_OK;
        }
        PG_HIP(hipMemsetAsync(t->d8r + t->rows * (size_t)128, 0, 64 * (size_t)128, ctx->stream));
    }
    t->s8r = t->s8 / 254.0f;
    PG_HIP(hipMemsetAsync(d_st, 0, 4, ctx->stream));
    table_quant8r_kernel<<<(uint32_t)ctx->num_cus * 16, 256, 0, ctx->stream>>>(t->d, t->rows, t->s8, t->s8r, t->d8r, d_st);
    PG_HIP(hipGetLastError());
    PG_HIP(hipMemcpyAsync(ctx->h_status + 340, d_st, 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    float r2;
    memcpy(&r2, ctx->h_status + 340, 4);
    // (accumulated in fp32 from fp32 residuals: a relative 1e-2 and an absolute 1e-7 N cover that many times over)
    t->resid2 = sqrtf(r2) * 1.01f + 1e-7f * t->max_norm;
    if (ctx->knobs.debug_scan)
        fprintf(stderr, "[pg] residual shadow: s8r %.4g, max second residual %.4g (first: %.4g)\n", t->s8r, t->resid2, t->resid8);
    t->r2_ok = true;
    return PG_OK;
}

int rescreen16_prep_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs, uint32_t nq) {
    rescreen16_prep_kernel<<<(nq + 63) / 64, 256, 0, ctx->stream>>>(rs.qpad, t->max_norm, t->resid2, rs.q16);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// rs.susp ([nq][rs.cap], counts rs.susp_cnt) -> rs.susp2 ([nq][rs.cap], counts rs.susp2w_cnt, zeroed here)
int rescreen16_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs, uint32_t nq) {
    PG_HIP(hipMemsetAsync(rs.susp2w_cnt, 0, sizeof(uint32_t) * kMaxQueries, ctx->stream));
    const double per_q = t->wide_susp > 0.0f ? (double)t->wide_susp : 50000.0;
    uint32_t blocks = (uint32_t)(per_q / 64.0 / 10.0 / 4.0) + 1;
    if (blocks > 2048u / nq) blocks = 2048u / nq;
    if (blocks < 4) blocks = 4;
    rescreen16_kernel<<<dim3(blocks, nq), 256, 0, ctx->stream>>>(t->d8, t->d8r, t->s8, t->s8r, rs.q16, rs.thr, rs.susp, rs.susp_cnt, rs.cap,
                                                                (uint32_t)t->rows, rs.susp2, rs.susp2w_cnt, rs.cap, rs.overflow);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // namespace pg
