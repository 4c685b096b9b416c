// rank_x3.hip — DNN3 in PG_PREC_BF16X3 ("split bf16"): the fp32 specification on the bf16 matrix pipe.
//
// Why the mode exists: the reference hands model outputs on as fp32 widened to f64 (algorithm/eas/easyrec_response.go:479-483,
// eas/tf_response.go:55-59) and north_star asks for scores within 1e-5 of that path.  PG_PREC_BF16 misses it (4e-5), the fp32
// MFMA meets it at 1/16 of the bf16 rate.  Here every operand of the two matrix layers is a pair of bf16 values, x = hi + lo,
// and a term is three products — lo_w·hi_x, hi_w·lo_x, hi_w·hi_x — into the fp32 accumulator: 2^-16 relative per product,
// scores within ~1e-7 of PG_PREC_F32's, three times the MFMA work of the bf16 mode.
//
// dnn3_x3_kernel: one persistent workgroup per CU over 128-item tiles, EIGHT waves in TWO ROLES that share each SIMD:
//   * waves 0-3 ("layer-1 waves", one per SIMD): gather the tile's table rows (a tile ahead, straight to registers), split
//     them into the hi / lo X tiles, and run layer 1 in chunks of 64 hidden columns — wave (mp, nb) owns item blocks 2mp,
//     2mp + 1 and column block nb of the chunk (48 MFMAs) — then relu, split, and store the chunk into a double-buffered
//     hi / lo LDS tile.  They also finish the previous tile's scores (four partials per item and head, sigmoid, store).
//   * waves 4-7 ("layer-2 waves"): wave wn keeps the fp32 accumulators of ALL 128 items x its H2 / 4 output columns for the
//     tile (128 registers at H2 = 256) and adds one chunk's 64-deep partial product per interval (96 MFMAs); at the end of
//     the tile: relu → dot with every head's w3 from the accumulators → one partial per (item, head, wave).
//   One barrier per chunk.  The two waves of a SIMD run different code between the same barriers, so one's LDS reads, global
//   loads and conversions sit under the other's MFMAs without any hand-made interleaving — and the matrix pipe sees
//   48 + 96 MFMAs per SIMD and interval whichever wave issues them.
// Where it stands (round 5, PG_X3_PROFILE-style cycle stamps and ablation builds on the dev library): the matrix pipe is 61 %
// busy at the 1.95 GHz the chip holds under this kernel.  An interval is 6.3-6.7 K cycles for 4.6 K of MFMA issue per SIMD;
// the layer-1 wave is its critical path (48 MFMAs + conversion: 2.9 K cycles with the pipe to itself, 5.4-6.8 K beside the
// layer-2 wave's 96 MFMAs).  Tried and dropped, all bit-identical, none faster than 1.18-1.20 ms: the conversion of chunk
// c under the MFMAs of chunk c + 1 (second accumulator set, layer-2 waves two intervals behind; as a block per k-step, and
// cut into pieces between the individual MFMAs), weight fragments re-requested per k-step, weight fragments two k-steps
// ahead in the layer-2 waves, priority to the layer-2 waves (-3 %).
// Round 6 (same box, `scripts/dev/x3_time.py`, builds with `make WS_EXTRA=-DPG_X3_ABL=n`): 1.205 ms; X stored UNSPLIT (ablation 1 = the
// most a pre-split hi / lo shadow of the table rows could save): 1.23; H1 stored with NO relu / split at all (ablation 3): 1.19; X
// fragments two k-steps ahead: 1.205.  The conversions are not what the kernel waits for, and neither is the matrix pipe's schedule:
// rocm-smi beside a loop of this kernel reads 1 377-1 381 W of the 1 400 W package limit at 2.04-2.09 GHz of 2.4 (`scripts/dev/
// power_probe.sh`, bench.py's `power` object) — the kernel runs at the power limit, a busier pipe gets a lower clock.
// Weights (768 KB at 512-256: every matrix as hi and lo fragments) do not fit the CU: they stream from L2 once per tile,
// global → registers, each fragment a k-step (layer 2) or a chunk (layer 1) ahead of its use.  A layer-2 fragment feeds
// four item blocks (hi fragments twice): 6 / 3 MFMAs per 1-KiB load.  LDS: X hi / lo 64 KB + two H1 chunks hi / lo 64 KB +
// the request's layer-1 partial, b2, the heads' w3 and partials.
#include "rank_mlp.hpp"

namespace pg {

#define X3_MFMA(acc, b, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(x))
#define X3_READY2(a0, a1) asm volatile("s_nop 3" : "+v"(a0), "+v"(a1))
#define X3_DONE2(a0, a1) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a0), "+v"(a1))
#define X3_READY1(a0) asm volatile("s_nop 3" : "+v"(a0))
#define X3_DONE1(a0) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a0))

constexpr int kX3Items = 128;
constexpr int kX3CH = 64;

template <int H1, int H2>
constexpr size_t x3_lds_bytes(uint32_t n_out) {
    return (size_t)2 * kX3Items * kDIN * 2 + (size_t)4 * kX3Items * kX3CH * 2 +
           (size_t)(H1 + H2 + n_out * H2 + kMaxHeads + n_out * 4 * kX3Items) * 4;
}

// relu → split → 4 consecutive columns of one row of an H1 chunk tile (128-B rows, quads keyed by (row >> 1) & 7: see
// ls_store_h_quad in rank_rs.hip); the lo tile lies LO bytes on
__device__ __forceinline__ float x3_relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff()); }

template <int LO>
__device__ __forceinline__ void x3_store_h_quad(char* tile, int row, int col, float v0, float v1, float v2, float v3) {
    uint2 ph, pl;
#if defined(PG_X3_ABL) && PG_X3_ABL == 3          // ablation (wrong results): H1 stored without relu / split — the conversion's whole cost
    ph.x = __float_as_uint(v0); pl.x = __float_as_uint(v1); ph.y = __float_as_uint(v2); pl.y = __float_as_uint(v3);
#else
    split_bf16x2(x3_relu(v0), x3_relu(v1), ph.x, pl.x);
    split_bf16x2(x3_relu(v2), x3_relu(v3), ph.y, pl.y);
#endif
    char* const d = tile + row * 128 + ((((col >> 3) ^ ((row >> 1) & 7))) << 4) + (col & 7) * 2;
    *reinterpret_cast<uint2*>(d) = ph;
    *reinterpret_cast<uint2*>(d + LO) = pl;
}

#if defined(PG_X3_ABL) && PG_X3_ABL == 1
__device__ __forceinline__ void store_x_quad_raw(char* tile, int row, int c, float4 v, int lo_off) {
    char* const d = tile + row * 256 + ((((c >> 1) ^ (row & 15))) << 4) + (c & 1) * 8;
    *reinterpret_cast<uint2*>(d) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
    *reinterpret_cast<uint2*>(d + lo_off) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
}
#endif

template <int H1, int H2>
__global__ __launch_bounds__(512, 1) void dnn3_x3_kernel(MlpArgs a) {
    constexpr int M = kX3Items, CH = kX3CH, NCH = H1 / CH, KS1 = kDIN / 16, KS2 = H1 / 16, KSC = CH / 16, NB2 = H2 / 128;
    constexpr int X_B = M * kDIN * 2, HC_B = M * CH * 2;
    static_assert(H2 == 128 || H2 == 256, "four layer-2 waves x one or two 32-column blocks");
    static_assert(NCH >= 2 && KSC == 4, "chunks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XH = smem;                                  // X hi tile; the lo tile X_B on
    char* const HC = smem + 2 * X_B;                        // H1 chunk buffers: [2][hi | lo]
    float* const c1s = reinterpret_cast<float*>(smem + 2 * X_B + 4 * HC_B);
    float* const b2s = c1s + H1;
    const uint32_t n_out = a.n_out;
    float* const w3s = b2s + H2;                            // [n_out][H2]
    float* const b3s = w3s + n_out * H2;                    // [kMaxHeads]
    float* const hps = b3s + kMaxHeads;                     // head partials [n_out][4 waves][128 items]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_tiles = *a.n_tiles;
    const uint32_t t_begin = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    const uint32_t t_end = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);
    if (t_begin >= t_end) return;
    for (int i = tid; i < H2; i += 512) {
        for (uint32_t o = 0; o < n_out; ++o) w3s[o * H2 + i] = a.w3[o * H2 + i];
        b2s[i] = a.b2[i];
    }
    if (tid < (int)n_out) b3s[tid] = a.b3v[tid];

    if (wave < 4) {
        // =========================================== layer-1 waves ===========================================
        // the layer-1 wave's MFMAs first: its relu / split / store then runs under the other wave's MFMAs (+1 %; the other
        // way round costs 3 %)
        asm volatile("s_setprio 2");
        const int mp = wave & 1, nb1 = wave >> 1;
        const char* const w1h_base = reinterpret_cast<const char*>(a.w1p);
        const char* const w1l_base = reinterpret_cast<const char*>(a.w1p_lo);
        struct Tile { uint32_t req, item0, cnt; };
        auto load_desc = [&](uint32_t t) {
            Tile d{0, 0, 0};
            if (t < t_end) {
                d.req = (uint32_t)__builtin_amdgcn_readfirstlane(a.tile_req[t]);
                d.item0 = (uint32_t)__builtin_amdgcn_readfirstlane(a.tile_item0[t]);
                d.cnt = (uint32_t)__builtin_amdgcn_readfirstlane(a.tile_cnt[t]);
            }
            return d;
        };
        // gather: 4 adjacent lanes per item (64 contiguous bytes per instruction), two passes of 64 items
        float4 xq[2][8];
        auto gather = [&](const Tile& d) {
            if (d.cnt == 0) return;
            uint32_t t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint32_t item = p * 64 + (t_ >> 2);
                uint32_t row = a.cand_rows[d.item0 + (item < d.cnt ? item : d.cnt - 1)];
                row = row < a.tab_rows ? row : a.tab_rows - 1;
                const float4* src = reinterpret_cast<const float4*>(a.tab + (size_t)row * kDIN) + (t_ & 3);
#pragma unroll
                for (int j = 0; j < 8; ++j) xq[p][j] = src[4 * j];
            }
        };
        uint32_t c1_req = 0xffffffffu;
        auto write_x = [&](const Tile& d) {                 // X tile + the request's layer-1 partial (the X tile is idle)
            if (d.cnt == 0) return;
            uint32_t t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#if defined(PG_X3_ABL) && PG_X3_ABL == 1      // ablation (wrong results): X stored unsplit — what a pre-split shadow of the rows could save at most
                    store_x_quad_raw(XH, p * 64 + (t_ >> 2), 4 * j + (t_ & 3), xq[p][j], X_B);
#else
                    store_x_quad<2>(XH, p * 64 + (t_ >> 2), 4 * j + (t_ & 3), xq[p][j], X_B);
#endif
                }
            if (d.req != c1_req) {
                c1_req = d.req;
                for (int i = t_; i < H1; i += 256) c1s[i] = a.c1[(size_t)d.req * a.c1_stride + i];
            }
        };
        // a finished tile's scores: z = b3 + the four layer-2 waves' partials in wave order; thread (item, head parity)
        auto finalize = [&](const Tile& f) {
            uint32_t t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
            const uint32_t item = t_ & (M - 1);
            if (item < f.cnt)
                for (uint32_t o = t_ >> 7; o < n_out; o += 2) {
                    float z = b3s[o];
#pragma unroll
                    for (int s = 0; s < 4; ++s) z += hps[(o * 4 + s) * M + item];
                    a.out[(size_t)o * a.out_stride + f.item0 + item] = 1.0f / (1.0f + expf(-z));
                }
        };
        bf16x8 w1h[KS1], w1l[KS1];
        auto load_w1 = [&](int c) {                         // fragments of n-block c * 2 + nb1, every k-step, hi and lo
            const uint32_t off = (uint32_t)__builtin_amdgcn_readfirstlane((c * 2 + nb1) * KS1 * 1024);
            uint32_t l_ = threadIdx.x;
            asm volatile("" : "+v"(l_));
            const uint32_t lane_off = (l_ & 63) * 16;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                w1h[ks] = *reinterpret_cast<const bf16x8*>(w1h_base + off + ks * 1024 + lane_off);
                w1l[ks] = *reinterpret_cast<const bf16x8*>(w1l_base + off + ks * 1024 + lane_off);
            }
        };

        Tile cur = load_desc(t_begin), fin{0, 0, 0};
        gather(cur);
        load_w1(0);
        write_x(cur);
        __syncthreads();                                    // prologue barrier
        for (uint32_t tile = t_begin; tile < t_end; ++tile) {
            const Tile nxt = load_desc(tile + 1);
            gather(nxt);                                    // lands during the tile, stored behind its last chunk
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                if (c == 1 && fin.cnt) finalize(fin);       // (its partials were written during interval 0)
                uint32_t t_ = threadIdx.x;
                asm volatile("" : "+v"(t_));
                const int i32 = t_ & 31, h = (t_ >> 5) & 1;
                f32x16 acc[2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 cv = *reinterpret_cast<const float4*>(c1s + c * CH + nb1 * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        acc[mb][4 * g + 0] = cv.x;
                        acc[mb][4 * g + 1] = cv.y;
                        acc[mb][4 * g + 2] = cv.z;
                        acc[mb][4 * g + 3] = cv.w;
                    }
                }
                X3_READY2(acc[0], acc[1]);
                const char* const xr0 = XH + ((2 * mp) * 32 + i32) * 256;
                const char* const xr1 = xr0 + 32 * 256;
                bf16x8 xh[2][2], xl[2][2];
                auto xfrag = [&](int ks, int s) {
                    const int q = ((ks * 2 + h) ^ (i32 & 15)) << 4;
                    xh[s][0] = *reinterpret_cast<const bf16x8*>(xr0 + q);
                    xh[s][1] = *reinterpret_cast<const bf16x8*>(xr1 + q);
                    xl[s][0] = *reinterpret_cast<const bf16x8*>(xr0 + X_B + q);
                    xl[s][1] = *reinterpret_cast<const bf16x8*>(xr1 + X_B + q);
                };
                xfrag(0, 0);
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    if (ks + 1 < KS1) xfrag(ks + 1, (ks + 1) & 1);
                    X3_MFMA(acc[0], w1l[ks], xh[ks & 1][0]);
                    X3_MFMA(acc[1], w1l[ks], xh[ks & 1][1]);
                    X3_MFMA(acc[0], w1h[ks], xl[ks & 1][0]);
                    X3_MFMA(acc[1], w1h[ks], xl[ks & 1][1]);
                    X3_MFMA(acc[0], w1h[ks], xh[ks & 1][0]);
                    X3_MFMA(acc[1], w1h[ks], xh[ks & 1][1]);
                }
                load_w1(c + 1 < NCH ? c + 1 : 0);           // next chunk's (next tile's first) fragments
                X3_DONE2(acc[0], acc[1]);
                char* const hb = HC + (c & 1) * (2 * HC_B);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        x3_store_h_quad<HC_B>(hb, (2 * mp + mb) * 32 + i32, nb1 * 32 + 8 * g + 4 * h, acc[mb][4 * g + 0],
                                              acc[mb][4 * g + 1], acc[mb][4 * g + 2], acc[mb][4 * g + 3]);
                __syncthreads();
            }
            write_x(nxt);                                   // every layer-1 wave is past its last read of this tile's X
            __syncthreads();
            fin = cur;
            cur = nxt;
        }
        __syncthreads();                                    // the layer-2 waves' head of the last tile
        finalize(fin);
    } else {
        // =========================================== layer-2 waves ===========================================
        const int wn = wave - 4;
        const char* const w2h_base = reinterpret_cast<const char*>(a.w2p) + (size_t)(wn * NB2) * KS2 * 1024;
        const char* const w2l_base = reinterpret_cast<const char*>(a.w2p_lo) + (size_t)(wn * NB2) * KS2 * 1024;
        f32x16 acc2[4][NB2];
        bf16x8 bh[2][NB2], bl[2][NB2];
        auto load_w2 = [&](int kk, int s) {
            uint32_t l_ = threadIdx.x;
            asm volatile("" : "+v"(l_));
            const uint32_t lane_off = (l_ & 63) * 16;
#pragma unroll
            for (int nb = 0; nb < NB2; ++nb) {
                bh[s][nb] = *reinterpret_cast<const bf16x8*>(w2h_base + (uint32_t)((nb * KS2 + kk) * 1024) + lane_off);
                bl[s][nb] = *reinterpret_cast<const bf16x8*>(w2l_base + (uint32_t)((nb * KS2 + kk) * 1024) + lane_off);
            }
        };
        auto init_acc = [&]() {
            uint32_t t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
            const int h = (t_ >> 5) & 1;
#pragma unroll
            for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(b2s + (wn * NB2 + nb) * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        acc2[mb][nb][4 * g + 0] = bv.x;
                        acc2[mb][nb][4 * g + 1] = bv.y;
                        acc2[mb][nb][4 * g + 2] = bv.z;
                        acc2[mb][nb][4 * g + 3] = bv.w;
                    }
                }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                if constexpr (NB2 == 2) X3_READY2(acc2[mb][0], acc2[mb][1]);
                else X3_READY1(acc2[mb][0]);
            }
        };
        // relu → dot with every head's w3 over this wave's columns: one partial per (head, item); a lane owns 16 * NB2 of
        // its item's columns, lanes i and i + 32 the two column halves of a block
        auto head = [&]() {
            uint32_t t_ = threadIdx.x;
            asm volatile("" : "+v"(t_));
            const int i32 = t_ & 31, h = (t_ >> 5) & 1;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                if constexpr (NB2 == 2) X3_DONE2(acc2[mb][0], acc2[mb][1]);
                else X3_DONE1(acc2[mb][0]);
            }
            for (uint32_t o = 0; o < n_out; ++o) {
                float4 wv[NB2][4];
#pragma unroll
                for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        wv[nb][g] = *reinterpret_cast<const float4*>(w3s + o * H2 + (wn * NB2 + nb) * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    float p = 0.0f;
#pragma unroll
                    for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 0], 0.0f), wv[nb][g].x, p);
                            p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 1], 0.0f), wv[nb][g].y, p);
                            p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 2], 0.0f), wv[nb][g].z, p);
                            p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 3], 0.0f), wv[nb][g].w, p);
                        }
                    p += __shfl_xor(p, 32);
                    if (h == 0) hps[(o * 4 + wn) * M + mb * 32 + i32] = p;
                }
            }
        };

        load_w2(0, 0);
        __syncthreads();                                    // prologue barrier
        for (uint32_t tile = t_begin; tile < t_end; ++tile) {
            if (tile != t_begin) head();                    // the previous tile's, under this tile's first layer-1 chunk
            init_acc();
            __syncthreads();
#pragma unroll 1
            for (int c = 0; c < NCH; ++c) {
                uint32_t t_ = threadIdx.x;
                asm volatile("" : "+v"(t_));
                const int i32 = t_ & 31, h = (t_ >> 5) & 1;
                const char* const hr = HC + (c & 1) * (2 * HC_B) + i32 * 128;
                const int sw = (i32 >> 1) & 7;
                // A fragments (hi, lo) of step f = (k-step f / 4, item block f % 4): three ahead in four rotating slots
                bf16x8 ah[4], al[4];
                auto afrag = [&](int f) {
                    const char* const p = hr + (f & 3) * (32 * 128) + ((((f >> 2) * 2 + h) ^ sw) << 4);
                    ah[f & 3] = *reinterpret_cast<const bf16x8*>(p);
                    al[f & 3] = *reinterpret_cast<const bf16x8*>(p + HC_B);
                };
                afrag(0);
                afrag(1);
                afrag(2);
#pragma unroll
                for (int f = 0; f < 4 * KSC; ++f) {
                    const int ks = f >> 2, mb = f & 3;
                    if (mb == 0) {                          // the next k-step's weight fragments (of the next chunk / tile behind the last)
                        const int kn = c * KSC + ks + 1;
                        load_w2(kn < KS2 ? kn : 0, (ks + 1) & 1);
                    }
                    if (f + 3 < 4 * KSC) afrag(f + 3);
#pragma unroll
                    for (int nb = 0; nb < NB2; ++nb) X3_MFMA(acc2[mb][nb], bl[ks & 1][nb], ah[f & 3]);
#pragma unroll
                    for (int nb = 0; nb < NB2; ++nb) X3_MFMA(acc2[mb][nb], bh[ks & 1][nb], al[f & 3]);
#pragma unroll
                    for (int nb = 0; nb < NB2; ++nb) X3_MFMA(acc2[mb][nb], bh[ks & 1][nb], ah[f & 3]);
                }
                __syncthreads();
            }
        }
        head();
        __syncthreads();
    }
}

template <int H1, int H2>
static int launch_x3(pg_ctx* ctx, const MlpArgs& a) {
    const size_t lds = x3_lds_bytes<H1, H2>(a.n_out);
    int rc;
    if ((rc = ensure_dyn_lds(ctx, (const void*)dnn3_x3_kernel<H1, H2>, lds))) return rc;
    dnn3_x3_kernel<H1, H2><<<ctx->num_cus, 512, lds, ctx->stream>>>(a);
    return PG_OK;
}

bool dnn3_x3_shape(uint32_t h1, uint32_t h2) {
    return (h1 == 128 && h2 == 128) || (h1 == 256 && h2 == 128) || (h1 == 256 && h2 == 256) || (h1 == 512 && h2 == 256);
}

int launch_dnn3_x3(pg_ctx* ctx, uint32_t h1, uint32_t h2, const MlpArgs& a) {
    if (h1 == 512 && h2 == 256) return launch_x3<512, 256>(ctx, a);
    if (h1 == 256 && h2 == 256) return launch_x3<256, 256>(ctx, a);
    if (h1 == 256 && h2 == 128) return launch_x3<256, 128>(ctx, a);
    if (h1 == 128 && h2 == 128) return launch_x3<128, 128>(ctx, a);
    set_error("rank: no split-bf16 kernel for hidden widths %u-%u", h1, h2);
    return PG_ERR_UNSUPPORTED;
}

}  // namespace pg
