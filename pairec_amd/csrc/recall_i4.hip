// recall_i4.hip — the small-batch screen: the full table pass of a recall with at most kI4MaxQueries queries
// streams a 4-bit shadow of the rows (68 B per row instead of the int8 shadow's 128) and bounds every row·query
// score on the vector ALU (v_dot4_i32_i8); rows whose bound reaches the query's threshold are suspects for
// rescore_kernel, exactly like screen_kernel's.  A single request is HBM-bound on the shadow it streams
// (1.9 ms over the int8 shadow of 100M x 128), so halving the bytes is what shortens its latency; from a few
// dozen queries on the pass is MFMA work and the int8 screen stays.  Reference path: the same
// VectorRecall.GetCandidateItems as recall.hip (module/vector_recall.go in the reference; the search itself is the
// external faiss service's IndexFlatIP) — results are bit-identical, the screen only decides what is re-scored.
//
// Shadow: x^_i = s_r X_i, X_i in [-7, 7], ONE scale per row s_r = max_i |x_i| / 7; a row is 64 B of nibbles — byte b
// of dword w holds dims 8w+b (low nibble) and 8w+4+b (high nibble), each stored as X + 8 — plus {s_r, R_r} as two bf16 in
// one dword (68 B per row in all; two fp32 until round 3: 72 B): s_r is the row's scale ROUNDED UP to a bf16 value BEFORE
// the row is quantised with it (so the stored scale is exactly the one the nibbles mean), R_r >= ||x - x^|| the row's own
// MEASURED residual, rounded up to bf16.
// Bound, for the true score s = sum x_i q_i and the integer dot product I = sum X_i Q_i (Q = int8 query, scale s_q):
//     |s - s_r s_q I| <= ||x - x^|| ||q|| + ||x^|| ||q - q^||  <=  R_r ||q|| + H_r ||q - q^||,
//     H_r = min(7 sqrt(dim) s_r, N + R4) >= ||x^||     (|X_i| <= 7; N = max row norm, R4 = max_r R_r)
// so a row can reach thr only if
//     s_r s_q I + R_r B_q + H_r A_q >= thr,   B_q = ||q|| (1 + slack),   A_q = ||q - q^|| + slack (||q|| + ||q - q^||)
// (slack: the rounding of the specification's fmaf chain, <= 128 x 2^-24 ||x|| ||q||, and of this fp32 evaluation).
// Every term is relative to the ROW — scale, residual, norm bound — so rows of very different magnitude (heavy-tailed
// norms, the tables whose range defeats the int8 shadow's single scale) screen as well as uniform ones.
#include "common.hpp"

namespace pg {

namespace {

constexpr int kStage4 = 128;          // staged suspects per wave and query (a load adds <= 16)
constexpr uint32_t kRescore4Blocks = 512;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Screen4Args {
    const u32x4* d4;          // [rows + 64][4] quads
    const uint32_t* d4s;      // [rows + 64] row scale (bf16, low half) | row residual bound (bf16, high half)
    const uint32_t* q4;       // [4][32] int8 queries, then [4][4] {s_q, B_q, A_q, 8 sum(Q) as int bits}
    float h_cap;              // N + R4
    const float* thr;         // [>= 4] running thresholds
    uint32_t* susp_cnt;
    uint32_t* susp;           // [4][cap4]
    uint32_t* overflow;
    uint32_t cap4, rows, row_begin;     // rows [row_begin, rows), row_begin a multiple of 64
    // squared-Euclidean recall (L2): a row is a suspect iff fmaf(2, f, -(|x|^2 + |q|^2)) >= thr - 2 delta, f = the bound of the
    // inner product above (thr in -d units; delta: the rounding of the specification's and of this evaluation)
    const float* nx;          // [rows + 64] |x|^2 of every row
    const float* l2c;         // [4][2] {|q|^2, 2 delta}
};

template <int NQ, bool L2 = false>
__global__ __launch_bounds__(256) void screen4_kernel(Screen4Args a) {
    __shared__ uint32_t stage[4][NQ][kStage4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int part = lane & 3;                         // which 32 dims of the row this lane holds
    int Q[NQ][8];
    float sq[NQ], bq[NQ], aq[NQ], tq[NQ], nqc[NQ];
    int bias[NQ];
    uint32_t cnt[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
        for (int j = 0; j < 8; ++j) Q[qi][j] = (int)a.q4[qi * 32 + part * 8 + j];
        const float* c = reinterpret_cast<const float*>(a.q4 + 4 * 32) + qi * 4;
        sq[qi] = c[0];
        bq[qi] = c[1];
        aq[qi] = c[2];
        const float t = a.thr[qi];
        bias[qi] = __float_as_int(c[3]);
        // anything not finite: every row is a suspect (the lists overflow, the next plan runs)
        tq[qi] = (t == t && aq[qi] == aq[qi] && aq[qi] < 1e30f && t > -__builtin_inff()) ? t : -__builtin_inff();
        nqc[qi] = 0.0f;
        if constexpr (L2) {
            nqc[qi] = a.l2c[2 * qi];
            const float d2 = a.l2c[2 * qi + 1];
            tq[qi] = (tq[qi] > -__builtin_inff() && d2 == d2 && d2 < 1e30f && nqc[qi] == nqc[qi]) ? tq[qi] - d2 : -__builtin_inff();
        }
        cnt[qi] = 0;
    }
    auto flush = [&](int qi) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.susp_cnt[qi], cnt[qi]);
        base = __builtin_amdgcn_readfirstlane(base);
        for (uint32_t i = lane; i < cnt[qi]; i += 64) {
            const uint32_t pos = base + i;
            if (pos < a.cap4) a.susp[(uint64_t)qi * a.cap4 + pos] = stage[w][qi][i];
            else *a.overflow = 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cnt[qi] = 0;
    };
    const uint32_t wave = blockIdx.x * 4 + w, nwaves = gridDim.x * 4;
    const uint32_t ngroups = (a.rows - a.row_begin + 63) / 64;   // 64 rows (4 KiB of shadow) per wave step, 4 loads in flight
    for (uint32_t g = wave; g < ngroups; g += nwaves) {
        const uint32_t row0 = a.row_begin + g * 64 + (lane >> 2);
        u32x4 v[4];
        uint32_t sr[4];
        f32x2 s[4];
        float nxv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = __builtin_nontemporal_load(a.d4 + (size_t)(row0 + 16 * u) * 4 + part);
            sr[u] = __builtin_nontemporal_load(a.d4s + row0 + 16 * u);
            if constexpr (L2) nxv[u] = __builtin_nontemporal_load(a.nx + row0 + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[u].x = __uint_as_float(sr[u] << 16);
            s[u].y = __uint_as_float(sr[u] & 0xffff0000u);
        }
        float hr[4];                                   // H_r (7 sqrt(128) = 79.196 rounded up)
#pragma unroll
        for (int u = 0; u < 4; ++u) hr[u] = fminf(s[u].x * 79.1961f, a.h_cap);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t row = row0 + 16 * u;
            const uint32_t d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            int b[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                b[2 * i] = (int)(d[i] & 0x0F0F0F0Fu);
                b[2 * i + 1] = (int)((d[i] >> 4) & 0x0F0F0F0Fu);
            }
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                int acc = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_sdot4(b[i], Q[qi][i], acc, false);
                acc += __builtin_amdgcn_update_dpp(0, acc, 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]
                acc += __builtin_amdgcn_update_dpp(0, acc, 0x4E, 0xF, 0xF, false);     // quad_perm [2,3,0,1]
                float f = __fmaf_rn(s[u].x * sq[qi], (float)(acc - bias[qi]), __fmaf_rn(hr[u], aq[qi], s[u].y * bq[qi]));
                if constexpr (L2) f = __fmaf_rn(2.0f, f, -(nxv[u] + nqc[qi]));
                const bool hit = part == 0 && row < a.rows && !(f < tq[qi]);
                const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
                if (m) {
                    const uint32_t n = (uint32_t)__popcll(m);
                    if (cnt[qi] + n > (uint32_t)kStage4) flush(qi);
                    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (hit) stage[w][qi][cnt[qi] + before] = row;
                    cnt[qi] += n;
                }
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
        if (cnt[qi]) flush(qi);
}

// per call: the int8 queries in plain order, their scales and the bound's per-query constants (4 queries, one wave each)
__global__ __launch_bounds__(256) void screen4_prep_kernel(const float* __restrict__ qpad, float max_norm, float rmax4,
                                                           uint32_t* __restrict__ q4) {
    const int qi = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* q = qpad + (size_t)qi * 128;
    const float v0 = q[lane], v1 = q[lane + 64];
    int bad = (!(fabsf(v0) <= 3.0e38f) || !(fabsf(v1) <= 3.0e38f)) ? 1 : 0;
    float mx = fmaxf(fabsf(v0), fabsf(v1));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        bad |= __shfl_xor(bad, off, 64);
    }
    const float sc = fmaxf(mx / 127.0f, 1e-30f);
    int Q0 = __float2int_rn(v0 / sc), Q1 = __float2int_rn(v1 / sc);
    Q0 = Q0 > 127 ? 127 : (Q0 < -127 ? -127 : Q0);
    Q1 = Q1 > 127 ? 127 : (Q1 < -127 ? -127 : Q1);
    if (bad) Q0 = Q1 = 0;
    const double d0 = (double)v0 - (double)sc * (double)Q0, d1 = (double)v1 - (double)sc * (double)Q1;
    double ss = (double)v0 * (double)v0 + (double)v1 * (double)v1, dd = d0 * d0 + d1 * d1;
    int sum = Q0 + Q1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off, 64);
        dd += __shfl_xor(dd, off, 64);
        sum += __shfl_xor(sum, off, 64);
    }
    int8_t* qb = reinterpret_cast<int8_t*>(q4) + qi * 128;
    qb[lane] = (int8_t)Q0;
    qb[lane + 64] = (int8_t)Q1;
    if (lane == 0) {
        const double nq = sqrt(ss), dq = sqrt(dd);
        const double N = (double)max_norm, R = (double)rmax4;
        // B: the factor of the row's residual R_r; A: the factor of the row's norm bound H_r >= ||x^||.  Slack: the
        // specification's fp32 fmaf chain rounds by <= 128 x 2^-24 ||x|| ||q|| <= 1e-5 (H_r + R_r) ||q||, this kernel's own
        // fp32 evaluation (five roundings of values below 2 (H_r + R_r)(||q|| + dq)) by <= 2e-6 (H_r + R_r)(||q|| + dq)
        (void)N; (void)R;
        const double C = nq * (1.0 + 1e-5 + 2e-6) * 1.000001 + 2e-6 * dq + 1e-30;
        const double E = dq * 1.0001 + (1e-5 + 2e-6) * nq + 2e-6 * dq + 1e-30;
        float* c = reinterpret_cast<float*>(q4 + 4 * 32) + qi * 4;
        c[0] = sc;
        c[1] = (float)(C * 1.000001);
        c[2] = bad ? __builtin_nanf("") : (float)(E * 1.000001);
        c[3] = __int_as_float(8 * sum);
    }
}

// the 4-bit shadow, its row scales and the measured constants of the bound.  A thread converts 8 consecutive values
// (one dword of shadow); 16 neighbouring lanes share a row.
// the smallest bf16 value >= f (f finite, >= 0), as fp32
__device__ __forceinline__ float bf16_up(float f) {
    const uint32_t b = __float_as_uint(f);
    return __uint_as_float((b & 0xffffu) ? (b | 0xffffu) + 1u : b);
}
__global__ __launch_bounds__(256) void table_quant4_kernel(const float* __restrict__ tab, uint64_t rows,
                                                           uint32_t* __restrict__ out4, uint32_t* __restrict__ out_scale,
                                                           float* __restrict__ out_stats) {
    constexpr int DIM = 128, G = 16;
    __shared__ float s_rho[4], s_r[4], s_lam[4];
    const uint64_t n8 = rows * (uint64_t)G;
    float rho_mx = 0.0f, r_mx = 0.0f, lam = 0.0f;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ((n8 + 63) & ~63ull);
         g += (uint64_t)gridDim.x * blockDim.x) {
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (g < n8) {
            const float4 a = reinterpret_cast<const float4*>(tab)[2 * g];
            const float4 b = reinterpret_cast<const float4*>(tab)[2 * g + 1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        float amax = 0.0f, ss = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            amax = fmaxf(amax, fabsf(v[i]));
            ss = __fmaf_rn(v[i], v[i], ss);
        }
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            amax = fmaxf(amax, __shfl_xor(amax, off, 64));
            ss += __shfl_xor(ss, off, 64);
        }
        // the stored scale IS the scale of the nibbles; never below 1e-30 (an all-zero row's nibbles mean zero under any scale, a
        // row of values below 7e-30 rounds to them and its residual says so): recall_i4m.hip divides by it
        const float s = bf16_up(fmaxf(amax / 7.0f, 1e-30f));
        const float inv = 1.0f / s;
        float rs = 0.0f;
        uint32_t word = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int X = __float2int_rn(v[i] * inv);
            X = X > 7 ? 7 : (X < -7 ? -7 : X);
            const float r = __fmaf_rn(-s, (float)X, v[i]);
            rs = __fmaf_rn(r, r, rs);
            word |= (uint32_t)(X + 8) << (8 * (i & 3) + 4 * (i >> 2));
        }
        if (g < n8) out4[g] = word;
#pragma unroll
        for (int off = 1; off < G; off <<= 1) rs += __shfl_xor(rs, off, 64);
        if (g < n8 && (g % G) == 0) {
            // (the residual was accumulated in fp32: a relative 1e-3 covers that, as for the int8 shadow)
            const float R = bf16_up(sqrtf(rs) * 1.001f + 1e-30f);
            out_scale[g / G] = (__float_as_uint(s) >> 16) | (__float_as_uint(R) & 0xffff0000u);
            if (ss > 0.0f) lam += R / sqrtf(ss);
            rho_mx = fmaxf(rho_mx, s > 1e-18f ? rs / (s * s * (float)DIM) : 0.0f);
            r_mx = fmaxf(r_mx, R);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        rho_mx = fmaxf(rho_mx, __shfl_xor(rho_mx, off, 64));
        r_mx = fmaxf(r_mx, __shfl_xor(r_mx, off, 64));
        lam += __shfl_xor(lam, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_rho[w] = rho_mx; s_r[w] = r_mx; s_lam[w] = lam; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(reinterpret_cast<uint32_t*>(out_stats), __float_as_uint(fmaxf(fmaxf(s_rho[0], s_rho[1]), fmaxf(s_rho[2], s_rho[3]))));
        atomicMax(reinterpret_cast<uint32_t*>(out_stats) + 1, __float_as_uint(fmaxf(fmaxf(s_r[0], s_r[1]), fmaxf(s_r[2], s_r[3]))));
        atomicAdd(out_stats + 2, s_lam[0] + s_lam[1] + s_lam[2] + s_lam[3]);   // (statistics only: decides whether the shadow is used)
    }
}

std::mutex g_i4_build_mu;      // a table is shared by the contexts of a device: one of them builds its shadow

}  // namespace

// the 4-bit shadow of a dim-128 table whose int8 statistics are valid (lazily, on the first small-batch recall)
int ensure_table_i4(pg_ctx* ctx, const pg_table* tc) {
    pg_table* t = const_cast<pg_table*>(tc);
    if (t->i4_ok || t->i4_failed) return PG_OK;
    std::lock_guard<std::mutex> g(g_i4_build_mu);
    if (t->i4_ok || t->i4_failed) return PG_OK;
    if (t->dim != 128 || !t->stats_valid || !t->all_finite) { t->i4_failed = true; return PG_OK; }
    void* p;
    int rc;
    if ((rc = scratch_reserve(ctx, 4, 4096, &p))) return rc;
    float* d_st = (float*)p + 320;
    if (!t->d4) {
        if (hipMalloc((void**)&t->d4, (t->rows + 64) * (size_t)64) != hipSuccess ||
            hipMalloc((void**)&t->d4s, (t->rows + 64) * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            if (t->d4) (void)hipFree(t->d4);
            t->d4 = nullptr;
            t->d4s = nullptr;
            t->i4_failed = true;                       // stay on the int8 screen
            return PG_OK;
        }
        PG_HIP(hipMemsetAsync(t->d4 + t->rows * (size_t)64, 0x88, 64 * (size_t)64, ctx->stream));
        PG_HIP(hipMemsetAsync(t->d4s + t->rows, 0, 64 * sizeof(uint32_t), ctx->stream));
    }
    PG_HIP(hipMemsetAsync(d_st, 0, 12, ctx->stream));
    table_quant4_kernel<<<(uint32_t)ctx->num_cus * 16, 256, 0, ctx->stream>>>(t->d, t->rows, (uint32_t*)t->d4, t->d4s, d_st);
    PG_HIP(hipGetLastError());
    PG_HIP(hipMemcpyAsync(ctx->h_status + 320, d_st, 12, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    float rho2, r2, lam;
    memcpy(&rho2, ctx->h_status + 320, 4);
    memcpy(&r2, ctx->h_status + 321, 4);
    memcpy(&lam, ctx->h_status + 322, 4);
    t->rho4 = sqrtf(rho2);                             // (diagnostic: 0.29 for a uniformly spread rounding error)
    t->rmax4 = r2 * 1.000001f + 1e-6f * t->max_norm;
    // mean over rows of the residual term in units of the score's spread ||x|| ||q|| / sqrt(dim): how far below the
    // K-th score the 4-bit bound reaches.  Uniform rows: 0.8, Gaussian rows: 1.3 (0.2 % / 0.5 % of the rows become
    // suspects at K / rows = 5e-5); beyond ~1.7 the re-scoring gathers more than the narrower shadow saves.
    t->lam4 = t->rows ? lam / (float)t->rows * sqrtf((float)t->dim) : 0.0f;
    if (ctx->knobs.debug_scan)
        fprintf(stderr, "[pg] 4-bit shadow: lambda %.3f, max rho %.3f, max residual %.4g (max norm %.4g)\n", t->lam4, t->rho4, t->rmax4, t->max_norm);
    t->i4_ok = true;
    return PG_OK;
}

int screen4_prep_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs) {
    screen4_prep_kernel<<<1, 256, 0, ctx->stream>>>(rs.qpad, t->max_norm, t->rmax4, rs.q4);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

uint32_t screen4_rescore_blocks() { return kRescore4Blocks; }

// squared-Euclidean recall: |q|^2 (the chain value the recall job computed) and twice the evaluation slack, per query
__global__ void screen4_l2_consts_kernel(const float* __restrict__ nqv, float max_norm, float* __restrict__ out) {
    const uint32_t q = threadIdx.x;
    if (q >= (uint32_t)kI4MaxQueries) return;
    const double nq = (double)nqv[q], N = (double)max_norm;
    out[2 * q] = nqv[q];
    out[2 * q + 1] = (float)((4e-6 * (N * N + nq + 2.0 * N * sqrt(nq)) + 1e-30) * 1.000001);
}

int screen4_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs, uint32_t nq, uint32_t row_begin, uint32_t rows,
                   uint32_t cap4, const float* l2_nqv) {
    Screen4Args a;
    a.nx = nullptr;
    a.l2c = nullptr;
    if (l2_nqv) {
        float* l2c = reinterpret_cast<float*>(rs.q4 + 4 * 32 + 16);       // behind the queries and their constants
        screen4_l2_consts_kernel<<<1, 64, 0, ctx->stream>>>(l2_nqv, t->max_norm, l2c);
        PG_HIP(hipGetLastError());
        a.nx = t->d_nx;
        a.l2c = l2c;
    }
    a.d4 = reinterpret_cast<const u32x4*>(t->d4);
    a.d4s = t->d4s;
    a.q4 = rs.q4;
    a.thr = rs.thr;
    a.susp_cnt = rs.susp_cnt;
    a.susp = rs.susp;
    a.overflow = rs.overflow;
    a.cap4 = cap4;
    a.rows = rows;
    a.row_begin = row_begin;
    a.h_cap = (t->max_norm + t->rmax4) * 1.000001f;
    // a persistent grid of exactly the resident workgroups (the group walk is interleaved over all waves, so waves
    // that started late would leave a tail); 4 KiB of loads in flight per wave
    static int per_cu[kI4MaxQueries + 1] = {0, 0, 0, 0, 0};
    const uint32_t n = nq < 1 ? 1 : (nq > kI4MaxQueries ? kI4MaxQueries : nq);
    if (!per_cu[n]) {
        int b = 0;
        const void* f = n == 1 ? (const void*)screen4_kernel<1> : n == 2 ? (const void*)screen4_kernel<2>
                      : n == 3 ? (const void*)screen4_kernel<3> : (const void*)screen4_kernel<4>;
        PG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, f, 256, 0));
        per_cu[n] = b < 1 ? 1 : (b > 8 ? 8 : b);
    }
    const uint32_t groups = (rows - row_begin + 63) / 64;
    uint32_t grid = (uint32_t)ctx->num_cus * (uint32_t)per_cu[n];
    if (grid > (groups + 3) / 4) grid = (groups + 3) / 4;
    if (l2_nqv) {
        switch (n) {
            case 1: screen4_kernel<1, true><<<grid, 256, 0, ctx->stream>>>(a); break;
            case 2: screen4_kernel<2, true><<<grid, 256, 0, ctx->stream>>>(a); break;
            case 3: screen4_kernel<3, true><<<grid, 256, 0, ctx->stream>>>(a); break;
            default: screen4_kernel<4, true><<<grid, 256, 0, ctx->stream>>>(a); break;
        }
        PG_HIP(hipGetLastError());
        return PG_OK;
    }
    switch (n) {
        case 1: screen4_kernel<1><<<grid, 256, 0, ctx->stream>>>(a); break;
        case 2: screen4_kernel<2><<<grid, 256, 0, ctx->stream>>>(a); break;
        case 3: screen4_kernel<3><<<grid, 256, 0, ctx->stream>>>(a); break;
        default: screen4_kernel<4><<<grid, 256, 0, ctx->stream>>>(a); break;
    }
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // namespace pg
