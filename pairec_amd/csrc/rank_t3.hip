// rank_t3.hip — DNN3 512-256 in bf16 with the weights stationary in the registers of TWELVE waves (three per SIMD).
//
// dnn3_ws_kernel (rank_ws.hip) runs ONE wave per SIMD: a wave issues in order, so its relu / convert / store, its LDS
// waits and its head arithmetic stand between its MFMAs — the matrix pipe is 43 % busy (profiles/r4_*).  A second
// wave on the SIMD hides all of that (dnn3_x3_kernel: 61 % busy without tuning), but the bf16 model's 384 KB of
// fragments leave no registers for it at two waves of 256.  At THREE waves of 168 registers they do:
//   * waves 0-3 ("layer-1 waves"): wave i keeps W1's fragments of hidden columns [128 i, +128) — 4 n-blocks x 8 k-steps =
//     32 fragments = 128 registers — and turns the tile's X (32 items) into that quarter of H1: per n-block the request's
//     partial as C, 8 MFMAs, relu → bf16 → LDS;
//   * waves 4-11 ("layer-2 waves"): wave j keeps W2's fragments of output columns [32 j, +32) — 32 k-steps = 32 fragments =
//     128 registers — and runs the 32-MFMA chain of the tile's H1 against them, then relu → dot with every head's w3 from
//     its 16 accumulator registers → one partial per (item, head, wave).  They also gather: table rows two tiles ahead
//     into registers, the bf16 X tile one tile ahead into LDS.
//   A SIMD hosts one layer-1 wave and two layer-2 waves (workgroup waves go to SIMDs cyclically): 96 MFMAs per 32-item
//   tile and SIMD from three dependent chains, each wave's non-matrix work under the other two's MFMAs.  ONE barrier per
//   tile; X and H1 tiles are double-buffered (layer 1 of tile k runs beside layer 2 of tile k - 1), scores are finished
//   by the layer-1 waves an interval later.  No weight traffic at all after the prologue.
// STATUS (round 5): an experiment behind PG_RANK_T3 / pg_set_option("rank_t3"), NOT the default — it ties dnn3_ws_kernel
// (0.52-0.55 ms per 1.28 M items both).  Cycle stamps and ablations (scripts/dev/t3_variants.sh): an interval is ~7 K cycles
// for 3 K of MFMA issue per SIMD.  With 168 registers a wave has room for ONE 32-item accumulator block, so a tile is 32
// MFMAs per wave, and what surrounds them — the barrier, the accumulators' initial LDS reads, relu / convert / store, the
// head, the gather role (1.2 K cycles per tile at the issue of two HBM-missing loads: 0.09 ms of the launch; every
// candidate = row 0: 0.44 ms) — is as long as the chain itself; prefetch depth (1-3 fragments), priorities and the order
// of the gather role within the interval change nothing.  The same measurements say what one wave per SIMD costs
// dnn3_ws_kernel: scripts/micro/mfma_chain.hip — a lone wave issues an MFMA every 33 cycles of the counter, two waves one
// per 24.6, three one per 21.7 (dependent or independent accumulators alike).
// Arithmetic: PG_PREC_BF16's (operands rounded to bf16, fp32 accumulation, k ascending per layer; the head sums a lane's
// 16 columns, the two column halves of a block, then the eight waves' partials in wave order): inside the mode's 1e-5.
#include "rank_mlp.hpp"

namespace pg {

#define T3_MFMA(acc, b, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(x))
#define T3_READY(a0) asm volatile("s_nop 3" : "+v"(a0))
#define T3_DONE(a0) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a0))

constexpr int kT3W2Lds = 6;      // the last k-steps' W2 fragments of every layer-2 wave live in LDS: 168 registers hold 26 + the rest of the wave
constexpr size_t t3_lds_bytes(uint32_t n_out) {
    return (size_t)2 * kT3Items * kDIN * 2 + (size_t)2 * kT3Items * 512 * 2 + (size_t)8 * kT3W2Lds * 1024 + (size_t)4 * 4 * 1024 +
           (size_t)(512 + 256 + n_out * 256 + kMaxHeads + 2 * n_out * 8 * kT3Items) * 4;
}
__device__ __forceinline__ float t3_relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff()); }

__global__ __launch_bounds__(768, 1) void dnn3_t3_kernel(MlpArgs a) {
    constexpr int M = kT3Items, H1 = 512, H2 = 256, KS1 = kDIN / 16, KS2 = H1 / 16;
    constexpr int X_B = M * kDIN * 2, H_B = M * H1 * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;                                  // [2] bf16 X tiles, 256-B rows, quads keyed by row & 15
    char* const HT = smem + 2 * X_B;                        // [2] bf16 H1 tiles, 1-KiB rows, quads keyed by row & 15
    char* const W2L = smem + 2 * X_B + 2 * H_B;             // [8 waves][kT3W2Lds] W2 fragments (k-steps KS2 - kT3W2Lds ..)
    char* const W1L = W2L + 8 * kT3W2Lds * 1024;            // [4 waves][4 n-blocks] W1 fragments of the last k-step
    float* const c1s = reinterpret_cast<float*>(W1L + 4 * 4 * 1024);              // [512]: every layer-1 wave its own 128
    float* const b2s = c1s + H1;
    const uint32_t n_out = a.n_out;
    float* const w3s = b2s + H2;                            // [n_out][256]
    float* const b3s = w3s + n_out * H2;                    // [kMaxHeads]
    float* const hps = b3s + kMaxHeads;                     // [2][n_out][8 waves][32 items]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_tiles = *a.n_tiles;
    const uint32_t t_begin = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    const uint32_t t_end = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);
    if (t_begin >= t_end) return;
    const uint32_t T = t_end - t_begin;
    for (int i = tid; i < H2; i += 768) {
        for (uint32_t o = 0; o < n_out; ++o) w3s[o * H2 + i] = a.w3[o * H2 + i];
        b2s[i] = a.b2[i];
    }
    if (tid < (int)n_out) b3s[tid] = a.b3v[tid];
    // Tile descriptors and candidate row ids are requested an interval (or more) before they are used — with one barrier per
    // 3 000 cycles of MFMA work a dependent round trip at the top of the interval (descriptor → row id → table row: three of
    // them) is most of the interval: load_raw issues the loads, uniform() — an iteration later — makes them wave-uniform.
    struct Tile { uint32_t req, item0, cnt; };
    auto load_raw = [&](uint32_t k, bool with_req = true) {  // tile k of this workgroup (cnt = 0 past the end); no wait
        Tile d{0, 0, 0};
        if (k < T) {
            if (with_req) d.req = a.tile_req[t_begin + k];  // (the layer-2 waves never look at the request: a register)
            d.item0 = a.tile_item0[t_begin + k];
            d.cnt = a.tile_cnt[t_begin + k];
        }
        return d;
    };
    auto uniform = [](const Tile& d) {
        return Tile{(uint32_t)__builtin_amdgcn_readfirstlane(d.req), (uint32_t)__builtin_amdgcn_readfirstlane(d.item0),
                    (uint32_t)__builtin_amdgcn_readfirstlane(d.cnt)};
    };

#ifdef PG_T3_PROFILE
    // developer aid (make WS_EXTRA=-DPG_T3_PROFILE): cycles per wave and phase, printed by the launcher
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
#define T3_MARK(i) { const uint64_t tn = __builtin_readcyclecounter(); ph[i] += tn - tp; tp = tn; }
#else
#define T3_MARK(i)
#endif
    if (wave < 4) {
        // =========================================== layer-1 waves ===========================================
#ifndef T3_NOPRIO
        asm volatile("s_setprio 2");                        // their chain first: relu / convert / store then hides
#endif
        bf16x8 w1r[4][KS1 - 1];                             // (the last k-step's four fragments: LDS — registers)
        char* const w1l = W1L + wave * (4 * 1024) + (tid & 63) * 16;
        {
            const uint32_t lane_off = (tid & 63) * 16;
            const char* const wb = reinterpret_cast<const char*>(a.w1p) + (size_t)(wave * 4) * KS1 * 1024;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                for (int ks = 0; ks < KS1 - 1; ++ks) w1r[nb][ks] = *reinterpret_cast<const bf16x8*>(wb + (nb * KS1 + ks) * 1024 + lane_off);
                *reinterpret_cast<uint4*>(w1l + nb * 1024) = *reinterpret_cast<const uint4*>(wb + (nb * KS1 + KS1 - 1) * 1024 + lane_off);
            }
        }
        // a finished tile's scores: z = b3 + the eight layer-2 waves' partials in wave order; thread (item, head group)
        auto finalize = [&](const Tile& f, uint32_t k) {
            uint32_t t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
            const uint32_t item = t_ & (M - 1);
            const float* const hp = hps + (size_t)(k & 1) * n_out * 8 * M;
            if (item < f.cnt)
                for (uint32_t o = t_ >> 5; o < n_out; o += 8) {
                    float z = b3s[o];
#pragma unroll
                    for (int s = 0; s < 8; ++s) z += hp[(o * 8 + s) * M + item];
                    a.out[(size_t)o * a.out_stride + f.item0 + item] = 1.0f / (1.0f + expf(-z));
                }
        };
        uint32_t c1_req = 0xffffffffu;
        Tile fin1{0, 0, 0}, fin2{0, 0, 0};                  // tiles k - 1 and k - 2
        Tile cur = uniform(load_raw(0));
        __syncthreads();                                    // prologue: X(0) is in LDS
        for (uint32_t k = 0; k < T + 2; ++k) {
            const Tile raw_next = load_raw(k + 1);
            T3_MARK(0)
            if (k >= 2 && fin2.cnt) finalize(fin2, k);      // (k - 2) & 1 == k & 1
            T3_MARK(1)
            if (cur.cnt) {
                uint32_t t_ = threadIdx.x;
                asm volatile("" : "+v"(t_));
                const int lane = t_ & 63, i32 = t_ & 31, h = (t_ >> 5) & 1;
                if (cur.req != c1_req) {                    // this wave's 128 columns of the request's layer-1 partial
                    c1_req = cur.req;
                    *reinterpret_cast<float2*>(c1s + wave * 128 + 2 * lane) =
                        *reinterpret_cast<const float2*>(a.c1 + (size_t)cur.req * a.c1_stride + wave * 128 + 2 * lane);
                }
                const char* const xr = XT + (k & 1) * X_B + i32 * 256;
                char* const hr = HT + (k & 1) * H_B + i32 * (H1 * 2);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    f32x16 acc;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 cv = *reinterpret_cast<const float4*>(c1s + wave * 128 + nb * 32 + 8 * g + 4 * h);
                        acc[4 * g + 0] = cv.x;
                        acc[4 * g + 1] = cv.y;
                        acc[4 * g + 2] = cv.z;
                        acc[4 * g + 3] = cv.w;
                    }
                    T3_READY(acc);
                    bf16x8 xf[2];
                    xf[0] = *reinterpret_cast<const bf16x8*>(xr + ((h ^ (i32 & 15)) << 4));
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        if (ks + 1 < KS1) xf[(ks + 1) & 1] = *reinterpret_cast<const bf16x8*>(xr + ((((ks + 1) * 2 + h) ^ (i32 & 15)) << 4));
                        if (ks < KS1 - 1) {
                            T3_MFMA(acc, w1r[nb][ks], xf[ks & 1]);
                        } else {
                            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(w1l + nb * 1024);
                            T3_MFMA(acc, wl, xf[ks & 1]);
                        }
                    }
                    T3_DONE(acc);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const f32x2 lo = {t3_relu(acc[4 * g + 0]), t3_relu(acc[4 * g + 1])};
                        const f32x2 hi = {t3_relu(acc[4 * g + 2]), t3_relu(acc[4 * g + 3])};
                        uint2 p;
                        p.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
                        p.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2));
                        const int col = wave * 128 + nb * 32 + 8 * g + 4 * h;
                        *reinterpret_cast<uint2*>(hr + (((col >> 3) ^ (i32 & 15)) << 4) + (col & 7) * 2) = p;
                    }
                }
            }
            T3_MARK(2)
            __syncthreads();
            T3_MARK(3)
            fin2 = fin1;
            fin1 = cur;
            cur = uniform(raw_next);
        }
    } else {
        // =========================================== layer-2 waves ===========================================
        const int wj = wave - 4;
        constexpr int KR = KS2 - kT3W2Lds;                   // k-steps whose fragments stay in registers
        bf16x8 w2r[KR];
        char* const w2l = W2L + wj * (kT3W2Lds * 1024) + (tid & 63) * 16;
        {
            const uint32_t lane_off = (tid & 63) * 16;
            const char* const wb = reinterpret_cast<const char*>(a.w2p) + (size_t)wj * KS2 * 1024;
#pragma unroll
            for (int ks = 0; ks < KR; ++ks) w2r[ks] = *reinterpret_cast<const bf16x8*>(wb + ks * 1024 + lane_off);
#pragma unroll
            for (int ks = KR; ks < KS2; ++ks)
                *reinterpret_cast<uint4*>(w2l + (ks - KR) * 1024) = *reinterpret_cast<const uint4*>(wb + ks * 1024 + lane_off);
        }
        // gather role: 512 threads, 16 per item, two 16-B quads each (quads q and q + 16 of the row's 32)
        const uint32_t gt = tid - 256, g_item = gt >> 4, g_q = gt & 15;
        float4 xq[2];
        auto load_rowid = [&](const Tile& d) -> uint32_t {   // (the value as stored: clamped where it is used)
            return d.cnt ? a.cand_rows[d.item0 + (g_item < d.cnt ? g_item : d.cnt - 1)] : 0u;
        };
        auto gather = [&](const Tile& d, uint32_t row) {
            if (d.cnt == 0) return;
            row = row < a.tab_rows ? row : a.tab_rows - 1;
#if defined(T3_ABL) && T3_ABL == 1               // (developer ablation: every candidate reads row 0 — no HBM misses; wrong results)
            row = 0;
#endif
            const float4* src = reinterpret_cast<const float4*>(a.tab + (size_t)row * kDIN) + g_q;
            xq[0] = src[0];
            xq[1] = src[16];
        };
        auto write_x = [&](const Tile& d, uint32_t k) {
            if (d.cnt == 0) return;
            store_x_quad<1>(XT + (k & 1) * X_B, (int)g_item, (int)g_q, xq[0]);
            store_x_quad<1>(XT + (k & 1) * X_B, (int)g_item, (int)g_q + 16, xq[1]);
        };
        Tile u1, u2, raw3;
        uint32_t rowid2;
        {
            const Tile d0 = uniform(load_raw(0));
            gather(d0, load_rowid(d0));
            write_x(d0, 0);
            u1 = uniform(load_raw(1, false));
            gather(u1, load_rowid(u1));
            u2 = uniform(load_raw(2, false));
            rowid2 = load_rowid(u2);
            raw3 = load_raw(3, false);
        }
        __syncthreads();                                    // prologue
        // The gather role of an interval: X of tile k + 1 (gathered during the previous interval) → LDS; rows of tile k + 2
        // requested (their ids came during the previous interval); the ids of tile k + 3 and the descriptor of tile k + 4
        // requested.  The two layer-2 waves of a SIMD do it at OPPOSITE ends of the interval (waves 4-7 before their MFMA
        // chain, waves 8-11 behind it): in step, both left the matrix pipe to the layer-1 wave for ~2 K cycles per tile and
        // then competed for it.
        Tile u3, raw4;
        uint32_t rowid3;
        auto gather_role = [&](uint32_t k) {
            T3_MARK(7)
            write_x(u1, k + 1);
            T3_MARK(4)
            gather(u2, rowid2);
            T3_MARK(5)
            u3 = uniform(raw3);
            rowid3 = load_rowid(u3);
            raw4 = load_raw(k + 4, false);
            T3_MARK(6)
        };
        auto layer2 = [&](uint32_t k) {                     // layer 2 + head of tile k - 1
            uint32_t t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
            const int i32 = t_ & 31, h = (t_ >> 5) & 1;
            const char* const hr = HT + ((k - 1) & 1) * H_B + i32 * (H1 * 2);
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(b2s + wj * 32 + 8 * g + 4 * h);
                acc[4 * g + 0] = bv.x;
                acc[4 * g + 1] = bv.y;
                acc[4 * g + 2] = bv.z;
                acc[4 * g + 3] = bv.w;
            }
            T3_READY(acc);
            // A fragments one k-step ahead, two slots: the wave has no register to spare (a spilled value's reload sits
            // behind the gather's HBM loads in the in-order vmcnt queue: 2 K cycles per tile), and with three waves on the
            // pipe an MFMA of this chain issues every ~100 cycles anyway
#ifndef T3_AHEAD
#define T3_AHEAD 1                                  // (2 and 3 spill 2 / 4 registers and are no faster)
#endif
            bf16x8 af[T3_AHEAD + 1];
#pragma unroll
            for (int s_ = 0; s_ < T3_AHEAD; ++s_) af[s_] = *reinterpret_cast<const bf16x8*>(hr + (((s_ * 2 + h) ^ (i32 & 15)) << 4));
            bf16x8 wl[2];                                   // the LDS-resident weight fragments: requested a k-step ahead as well
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                if (ks + T3_AHEAD < KS2)
                    af[(ks + T3_AHEAD) % (T3_AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(hr + ((((ks + T3_AHEAD) * 2 + h) ^ (i32 & 15)) << 4));
                if (ks + 1 >= KR && ks + 1 < KS2) wl[(ks + 1) & 1] = *reinterpret_cast<const bf16x8*>(w2l + (ks + 1 - KR) * 1024);
                if (ks < KR) {
                    T3_MFMA(acc, w2r[ks], af[ks % (T3_AHEAD + 1)]);
                } else {
                    T3_MFMA(acc, wl[ks & 1], af[ks % (T3_AHEAD + 1)]);
                }
            }
            T3_DONE(acc);
            T3_MARK(1)
            float* const hp = hps + (size_t)((k - 1) & 1) * n_out * 8 * M;
            for (uint32_t o = 0; o < n_out; ++o) {
                float p = 0.0f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 wv = *reinterpret_cast<const float4*>(w3s + o * H2 + wj * 32 + 8 * g + 4 * h);
                    p = __fmaf_rn(t3_relu(acc[4 * g + 0]), wv.x, p);
                    p = __fmaf_rn(t3_relu(acc[4 * g + 1]), wv.y, p);
                    p = __fmaf_rn(t3_relu(acc[4 * g + 2]), wv.z, p);
                    p = __fmaf_rn(t3_relu(acc[4 * g + 3]), wv.w, p);
                }
                p += __shfl_xor(p, 32);
                if (h == 0) hp[(o * 8 + wj) * M + i32] = p;
            }
            T3_MARK(2)
        };
        for (uint32_t k = 0; k < T + 2; ++k) {
            if (wj < 4) {
                gather_role(k);
                T3_MARK(0)
                if (k >= 1 && k <= T) layer2(k);
            } else {
                if (k >= 1 && k <= T) layer2(k);
                gather_role(k);
                T3_MARK(0)
            }
            __syncthreads();
            T3_MARK(3)
            u1 = u2;
            u2 = u3;
            raw3 = raw4;
            rowid2 = rowid3;
        }
    }
#ifdef PG_T3_PROFILE
    if ((tid & 63) == 0 && blockIdx.x < 2) {
        uint64_t* o_ = (uint64_t*)(a.field_emb) + (blockIdx.x * 12 + wave) * 8;
        for (int i = 0; i < 8; ++i) o_[i] = ph[i];
    }
#endif
}

int launch_dnn3_t3(pg_ctx* ctx, const MlpArgs& a) {
    const size_t lds = t3_lds_bytes(a.n_out);
    int rc;
    if ((rc = ensure_dyn_lds(ctx, (const void*)dnn3_t3_kernel, lds))) return rc;
#ifdef PG_T3_PROFILE
    static uint64_t* dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, 2 * 12 * 8 * 8);
    MlpArgs b = a;
    b.field_emb = reinterpret_cast<const float* const*>(dbg);
    dnn3_t3_kernel<<<ctx->num_cus, 768, lds, ctx->stream>>>(b);
    uint64_t hcyc[2 * 12 * 8];
    (void)hipMemcpy(hcyc, dbg, sizeof hcyc, hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls == 30) {
        fprintf(stderr, "t3 cycles per wave: L1 waves [loop top | finalize | tile work | barrier], L2 waves [write_x + requests | MFMA chain | head | barrier]\n");
        for (int wv = 0; wv < 12; ++wv) {
            fprintf(stderr, "  wg 1 wave %2d:", wv);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %9llu", (unsigned long long)hcyc[(12 + wv) * 8 + i]);
            fprintf(stderr, "\n");
        }
    }
    return PG_OK;
#endif
    dnn3_t3_kernel<<<ctx->num_cus, 768, lds, ctx->stream>>>(a);
    return PG_OK;
}

}  // namespace pg
