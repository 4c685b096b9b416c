// rank_rs.hip — the bf16 DNN3 rank kernel for the small hidden shapes: every weight resident in registers.
#include "rank_mlp.hpp"

namespace pg {

// ---------------------------------------------------------------------------------------------
// dnn3_rs_kernel<H1, H2>: DNN3 in bf16 for hidden widths 128-128, 256-128, 256-256 (EAS serves whatever the model
// is, algorithm/eas/model.go:197-222; service/rank/rank_service.go:264-289 calls it per request).
//
// At these widths an item costs 66-197 Kflop and a 512-B row gather: the stage is bound by the GATHER (1.28 M random
// rows of a 51-GB table), not by the matrix pipe, and mlp_kernel — which re-reads the weights from L2 for every tile
// and waits for its own tile's rows — spent 0.50-0.67 ms per 1.28 M items on them, as long as the 512-256 model takes.
// Here one persistent workgroup per CU (4 waves, one per SIMD) walks a contiguous range of 64-item tiles like
// dnn3_ws_kernel, but the whole model is stationary: wave w keeps the B fragments of its H1/4 hidden columns of layer 1
// (32-64 registers) and of its H2/4 output columns of layer 2 (32-128 registers) for the launch, so the tile loop issues
// no weight load at all and the only global traffic is the gather, which runs a tile ahead: descriptors three tiles
// ahead, candidate row ids two, the table rows of tile t+1 are requested as soon as tile t's have been written to the
// X tile.  Four adjacent lanes read 64 contiguous bytes of a row per instruction (16 rows per instruction).
// Arithmetic and k order of layers 1 and 2 are mlp_kernel<1, H1, H2, …>'s; the head sums a lane's H2/8 columns, then the
// item's 8 partials in slot order, as dnn3_ws_kernel does (a fixed order inside the bf16 mode's 1e-5).
// ---------------------------------------------------------------------------------------------
template <int H1, int H2>
constexpr size_t rs_lds_bytes() {
    return (size_t)kWsItems * (kDIN + H1) * 2 + (size_t)(H1 + 2 * H2 + 8 * kWsItems) * 4;
}

struct RsTile {
    uint32_t req, item0, cnt;
};

template <int H1, int H2>
__global__ __launch_bounds__(256, 1) void dnn3_rs_kernel(MlpArgs a) {
    constexpr int NB1 = H1 / 128, NB2 = H2 / 128, KS1 = kDIN / 16, KS2 = H1 / 16;
    constexpr int XT_B = kWsItems * kDIN * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;
    char* const H1T = smem + XT_B;
    float* const c1s = reinterpret_cast<float*>(smem + XT_B + (size_t)kWsItems * H1 * 2);
    float* const b2s = c1s + H1;
    float* const w3s = b2s + H2;
    float* const hps = w3s + H2;                                          // head partials [8 slots][64 items]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_tiles = *a.n_tiles;
    const uint32_t t_begin = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    const uint32_t t_end = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);
    if (t_begin >= t_end) return;

    // the model, for the whole launch
    bf16x8 w1r[NB1][KS1], w2r[NB2][KS2];
#pragma unroll
    for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
            w1r[nb][ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(a.w1p) +
                                                           (size_t)((wave * NB1 + nb) * KS1 + ks) * 1024 + lane * 16);
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
            w2r[nb][ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(a.w2p) +
                                                           (size_t)((wave * NB2 + nb) * KS2 + ks) * 1024 + lane * 16);
    if (tid < H2) {
        w3s[tid] = a.w3[tid];
        b2s[tid] = a.b2[tid];
    }

    // gather role: 4 adjacent lanes per item, lane l of them takes quads 4j + l (64 contiguous bytes per instruction)
    const int g_item = tid >> 2, g_l = tid & 3;
    auto load_desc = [&](uint32_t t) {
        RsTile d{0, 0, 0};
        if (t < t_end) {
            d.req = a.tile_req[t];
            d.item0 = a.tile_item0[t];
            d.cnt = a.tile_cnt[t];
        }
        return d;
    };
    auto uniform = [](const RsTile& d) {
        return RsTile{(uint32_t)__builtin_amdgcn_readfirstlane(d.req), (uint32_t)__builtin_amdgcn_readfirstlane(d.item0),
                      (uint32_t)__builtin_amdgcn_readfirstlane(d.cnt)};
    };
    auto load_rowid = [&](const RsTile& d) -> uint32_t {
        if (d.cnt == 0) return 0;
        return a.cand_rows[d.item0 + ((uint32_t)g_item < d.cnt ? (uint32_t)g_item : d.cnt - 1)];
    };
    float4 xq[8];
    auto load_rows = [&](uint32_t row) {
        row = row < a.tab_rows ? row : a.tab_rows - 1;
        const float4* src = reinterpret_cast<const float4*>(a.tab + (size_t)row * kDIN) + g_l;
#pragma unroll
        for (int j = 0; j < 8; ++j) xq[j] = src[4 * j];
    };
    RsTile cur = uniform(load_desc(t_begin)), nxt = uniform(load_desc(t_begin + 1)), nn = uniform(load_desc(t_begin + 2));
    load_rows(load_rowid(cur));
    uint32_t row_n1 = load_rowid(nxt);
    uint32_t c1_req = 0xffffffffu;
    const int i32 = lane & 31, h = lane >> 5, sw = lane & 15;
    const uint32_t fin_item = (uint32_t)wave * (kWsItems / 4) + (lane & 15);

    for (uint32_t tile = t_begin; tile < t_end; ++tile) {
        const RsTile d3 = load_desc(tile + 3);
        // ---- X tile from the rows requested a tile ago; then the next tile's rows and the row ids behind them
#pragma unroll
        for (int j = 0; j < 8; ++j) store_x_quad<1>(XT, g_item, 4 * j + g_l, xq[j]);
        load_rows(nxt.cnt ? row_n1 : 0);
        const uint32_t row_n2 = load_rowid(nn);
        if (cur.req != c1_req) {
            c1_req = cur.req;
            if (tid < H1) c1s[tid] = a.c1[(size_t)cur.req * a.c1_stride + tid];
        }
        __syncthreads();

        // ---- layer 1: hidden columns of n-blocks wave * NB1 + nb, transposed accumulators (a lane owns 4 consecutive
        // columns of one item)
        {
            const char* const x0 = XT + i32 * 256;
            const char* const x1 = XT + (32 + i32) * 256;
            f32x16 acc[2][NB1];
#pragma unroll
            for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 cv = *reinterpret_cast<const float4*>(c1s + (wave * NB1 + nb) * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        acc[mb][nb][4 * g + 0] = cv.x;
                        acc[mb][nb][4 * g + 1] = cv.y;
                        acc[mb][nb][4 * g + 2] = cv.z;
                        acc[mb][nb][4 * g + 3] = cv.w;
                    }
                }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(x0 + (((ks * 2 + h) ^ sw) << 4));
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(x1 + (((ks * 2 + h) ^ sw) << 4));
#pragma unroll
                for (int nb = 0; nb < NB1; ++nb) {
                    acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1r[nb][ks], a0, acc[0][nb], 0, 0, 0);
                    acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1r[nb][ks], a1, acc[1][nb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        store_h_quad<1, H1>(H1T, mb * 32 + i32, (wave * NB1 + nb) * 32 + 8 * g + 4 * h,
                                            fmaxf(acc[mb][nb][4 * g + 0], 0.0f), fmaxf(acc[mb][nb][4 * g + 1], 0.0f),
                                            fmaxf(acc[mb][nb][4 * g + 2], 0.0f), fmaxf(acc[mb][nb][4 * g + 3], 0.0f));
        }
        __syncthreads();

        // ---- layer 2 + head partials
        {
            const char* const h1r0 = H1T + i32 * (H1 * 2);
            const char* const h1r1 = H1T + (32 + i32) * (H1 * 2);
            f32x16 acc[2][NB2];
#pragma unroll
            for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(b2s + (wave * NB2 + nb) * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        acc[mb][nb][4 * g + 0] = bv.x;
                        acc[mb][nb][4 * g + 1] = bv.y;
                        acc[mb][nb][4 * g + 2] = bv.z;
                        acc[mb][nb][4 * g + 3] = bv.w;
                    }
                }
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(h1r0 + (((ks * 2 + h) ^ sw) << 4));
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(h1r1 + (((ks * 2 + h) ^ sw) << 4));
#pragma unroll
                for (int nb = 0; nb < NB2; ++nb) {
                    acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2r[nb][ks], a0, acc[0][nb], 0, 0, 0);
                    acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2r[nb][ks], a1, acc[1][nb], 0, 0, 0);
                }
            }
            // relu → dot head from the accumulators: the lane's partial runs over its columns in ascending order
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                float p = 0.0f;
#pragma unroll
                for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 wv = *reinterpret_cast<const float4*>(w3s + (wave * NB2 + nb) * 32 + 8 * g + 4 * h);
                        p = __fmaf_rn(fmaxf(acc[mb][nb][4 * g + 0], 0.0f), wv.x, p);
                        p = __fmaf_rn(fmaxf(acc[mb][nb][4 * g + 1], 0.0f), wv.y, p);
                        p = __fmaf_rn(fmaxf(acc[mb][nb][4 * g + 2], 0.0f), wv.z, p);
                        p = __fmaf_rn(fmaxf(acc[mb][nb][4 * g + 3], 0.0f), wv.w, p);
                    }
                hps[(wave * 2 + h) * kWsItems + mb * 32 + i32] = p;
            }
        }
        __syncthreads();                                   // partials visible; everyone is done with X and H1

        // ---- scores: z = (((b3 + p0) + p1) + …) + p7, 16 items per wave
        if (lane < kWsItems / 4 && fin_item < cur.cnt) {
            float z = a.b3;
#pragma unroll
            for (int s = 0; s < 8; ++s) z += hps[s * kWsItems + fin_item];
            a.out[cur.item0 + fin_item] = 1.0f / (1.0f + expf(-z));
        }
        cur = nxt;
        nxt = nn;
        nn = uniform(d3);
        row_n1 = row_n2;
    }
}

template <int H1, int H2>
static int launch_rs(pg_ctx* ctx, const MlpArgs& a) {
    constexpr size_t lds = rs_lds_bytes<H1, H2>();
    int rc;
    if ((rc = ensure_dyn_lds(ctx, (const void*)dnn3_rs_kernel<H1, H2>, lds))) return rc;
    dnn3_rs_kernel<H1, H2><<<ctx->num_cus, 256, lds, ctx->stream>>>(a);
    return PG_OK;
}

bool dnn3_rs_shape(uint32_t h1, uint32_t h2) {
    return (h1 == 128 && h2 == 128) || (h1 == 256 && h2 == 128) || (h1 == 256 && h2 == 256);
}

int launch_dnn3_rs(pg_ctx* ctx, uint32_t h1, uint32_t h2, const MlpArgs& a) {
    if (h1 == 128 && h2 == 128) return launch_rs<128, 128>(ctx, a);
    if (h1 == 256 && h2 == 128) return launch_rs<256, 128>(ctx, a);
    if (h1 == 256 && h2 == 256) return launch_rs<256, 256>(ctx, a);
    set_error("rank: no register-stationary kernel for hidden widths %u-%u", h1, h2);
    return PG_ERR_UNSUPPORTED;
}

}  // namespace pg
