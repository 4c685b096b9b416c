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
// (n_out heads on the shared trunk: w3 rows and head partials per head, then the heads' biases)
template <int H1, int H2, bool W1L>
constexpr size_t rs_lds_bytes(uint32_t n_out) {
    return (size_t)kWsItems * (kDIN + H1) * 2 + (size_t)(H1 + H2 + n_out * (H2 + 8 * kWsItems) + kMaxHeads) * 4 +
           (W1L ? (size_t)kDIN * H1 * 2 : 0);
}

// make WS_EXTRA=-DPG_RS_PROFILE: per-phase cycle counts of the first workgroups (developer aid)
#ifdef PG_RS_PROFILE
#define RS_MARK(i) { const uint64_t tn = __builtin_readcyclecounter(); ph[i] += tn - tp; tp = tn; }
#else
#define RS_MARK(i)
#endif

struct RsTile {
    uint32_t req, item0, cnt;
};

template <int H1, int H2, int WPC, int MBG, int MSPLIT, bool W1L>
__global__ __launch_bounds__(256 * MSPLIT, WPC) void dnn3_rs_kernel(MlpArgs a) {
    constexpr int NB1 = H1 / 128, NB2 = H2 / 128, KS1 = kDIN / 16, KS2 = H1 / 16;
    constexpr int MPW = 2 / MSPLIT;                        // 32-item blocks per wave
    constexpr int GL = 4 * MSPLIT, GQ = 32 / GL;           // gather: lanes per item, 16-B quads per lane
    constexpr int XT_B = kWsItems * kDIN * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;
    char* const H1T = smem + XT_B;
    float* const c1s = reinterpret_cast<float*>(smem + XT_B + (size_t)kWsItems * H1 * 2);
    float* const b2s = c1s + H1;
    const uint32_t n_out = a.n_out;                                       // heads on the shared trunk (1: the plain DNN3)
    float* const w3s = b2s + H2;                                          // [n_out][H2]
    float* const hps = w3s + n_out * H2;                                  // head partials [n_out][8 slots][64 items]
    float* const b3s = hps + n_out * 8 * kWsItems;                        // [kMaxHeads]
    char* const W1S = reinterpret_cast<char*>(b3s + kMaxHeads);           // W1L: layer 1's fragments, shared by the waves of a column slice
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id & 3;                          // column slice
    const int mbase = MSPLIT == 2 ? (wave_id >> 2) : 0;    // MSPLIT = 2: eight waves, waves w and w + 4 share a column slice and take one item block each
    const uint32_t n_tiles = *a.n_tiles;
    const uint32_t t_begin = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    const uint32_t t_end = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);
    if (t_begin >= t_end) return;

    // the model, for the whole launch
    bf16x8 w1r[W1L ? 1 : NB1][W1L ? 1 : KS1], w2r[NB2][KS2];
    if constexpr (W1L) {
        for (int i = tid; i < kDIN * H1 * 2 / 16; i += 256 * MSPLIT)
            reinterpret_cast<uint4*>(W1S)[i] = reinterpret_cast<const uint4*>(a.w1p)[i];
    } else {
#pragma unroll
        for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
                w1r[nb][ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(a.w1p) +
                                                               (size_t)((wave * NB1 + nb) * KS1 + ks) * 1024 + lane * 16);
    }
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
            w2r[nb][ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(a.w2p) +
                                                           (size_t)((wave * NB2 + nb) * KS2 + ks) * 1024 + lane * 16);
    if (tid < H2) {
        for (uint32_t o = 0; o < n_out; ++o) w3s[o * H2 + tid] = a.w3[o * H2 + tid];
        b2s[tid] = a.b2[tid];
    }
    if (tid < (int)n_out) b3s[tid] = a.b3v[tid];

    // gather role: 4 adjacent lanes per item, lane l of them takes quads 4j + l (64 contiguous bytes per instruction)
    const int g_item = tid / GL, g_l = tid % GL;
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
    float4 xq[GQ];
    auto load_rows = [&](uint32_t row) {
        row = row < a.tab_rows ? row : a.tab_rows - 1;
        const float4* src = reinterpret_cast<const float4*>(a.tab + (size_t)row * kDIN) + g_l;
#pragma unroll
        for (int j = 0; j < GQ; ++j) xq[j] = src[GL * j];
    };
    RsTile cur = uniform(load_desc(t_begin)), nxt = uniform(load_desc(t_begin + 1)), nn = uniform(load_desc(t_begin + 2));
    load_rows(load_rowid(cur));
    uint32_t row_n1 = load_rowid(nxt);
    uint32_t c1_req = 0xffffffffu;
    constexpr int FPW = kWsItems / (4 * MSPLIT);           // items a wave finishes
    const uint32_t fin_item = (uint32_t)wave_id * FPW + (lane & (FPW - 1));

#ifdef PG_RS_PROFILE
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
#endif
    for (uint32_t tile = t_begin; tile < t_end; ++tile) {
        const RsTile d3 = load_desc(tile + 3);
        // (per-lane addresses from an opaque copy of the thread id: hoisted out of the tile loop they are spilled)
        uint32_t tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int i32 = tid_o & 31, h = (tid_o >> 5) & 1, sw = tid_o & 15;
        // ---- X tile from the rows requested a tile ago; then the next tile's rows and the row ids behind them
#pragma unroll
        for (int j = 0; j < GQ; ++j) store_x_quad<1>(XT, g_item, GL * j + g_l, xq[j]);
        load_rows(nxt.cnt ? row_n1 : 0);
        const uint32_t row_n2 = load_rowid(nn);
        if (cur.req != c1_req) {
            c1_req = cur.req;
            if (tid < H1) c1s[tid] = a.c1[(size_t)cur.req * a.c1_stride + tid];
        }
        RS_MARK(0)
        __syncthreads();
        RS_MARK(1)

        // ---- layer 1: hidden columns of n-blocks wave * NB1 + nb, transposed accumulators (a lane owns 4 consecutive
        // columns of one item); MBG item blocks at a time (one where two workgroups share the CU's registers)
#pragma unroll 1
        for (int m0 = 0; m0 < MPW; m0 += MBG) {
            f32x16 acc[MBG][NB1];
#pragma unroll
            for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 cv = *reinterpret_cast<const float4*>(c1s + (wave * NB1 + nb) * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int mb = 0; mb < MBG; ++mb) {
                        acc[mb][nb][4 * g + 0] = cv.x;
                        acc[mb][nb][4 * g + 1] = cv.y;
                        acc[mb][nb][4 * g + 2] = cv.z;
                        acc[mb][nb][4 * g + 3] = cv.w;
                    }
                }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                bf16x8 af[MBG];
#pragma unroll
                for (int mb = 0; mb < MBG; ++mb)
                    af[mb] = *reinterpret_cast<const bf16x8*>(XT + ((mbase + m0 + mb) * 32 + i32) * 256 + (((ks * 2 + h) ^ sw) << 4));
#pragma unroll
                for (int nb = 0; nb < NB1; ++nb) {
                    bf16x8 bw;
                    if constexpr (W1L)
                        bw = *reinterpret_cast<const bf16x8*>(W1S + ((wave * NB1 + nb) * KS1 + ks) * 1024 + (tid_o & 63) * 16);
                    else
                        bw = w1r[nb][ks];
#pragma unroll
                    for (int mb = 0; mb < MBG; ++mb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, af[mb], acc[mb][nb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mb = 0; mb < MBG; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        store_h_quad<1, H1>(H1T, (mbase + m0 + mb) * 32 + i32, (wave * NB1 + nb) * 32 + 8 * g + 4 * h,
                                            fmaxf(acc[mb][nb][4 * g + 0], 0.0f), fmaxf(acc[mb][nb][4 * g + 1], 0.0f),
                                            fmaxf(acc[mb][nb][4 * g + 2], 0.0f), fmaxf(acc[mb][nb][4 * g + 3], 0.0f));
        }
        RS_MARK(2)
        __syncthreads();
        RS_MARK(3)

        // ---- layer 2 + head partials
#pragma unroll 1
        for (int m0 = 0; m0 < MPW; m0 += MBG) {
            f32x16 acc[MBG][NB2];
#pragma unroll
            for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(b2s + (wave * NB2 + nb) * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int mb = 0; mb < MBG; ++mb) {
                        acc[mb][nb][4 * g + 0] = bv.x;
                        acc[mb][nb][4 * g + 1] = bv.y;
                        acc[mb][nb][4 * g + 2] = bv.z;
                        acc[mb][nb][4 * g + 3] = bv.w;
                    }
                }
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                bf16x8 af[MBG];
#pragma unroll
                for (int mb = 0; mb < MBG; ++mb)
                    af[mb] = *reinterpret_cast<const bf16x8*>(H1T + ((mbase + m0 + mb) * 32 + i32) * (H1 * 2) + (((ks * 2 + h) ^ sw) << 4));
#pragma unroll
                for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
                    for (int mb = 0; mb < MBG; ++mb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2r[nb][ks], af[mb], acc[mb][nb], 0, 0, 0);
            }
            // relu → dot head from the accumulators: the lane's partial runs over its columns in ascending order
#pragma unroll
            for (int mb = 0; mb < MBG; ++mb) {
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
                hps[(wave * 2 + h) * kWsItems + (mbase + m0 + mb) * 32 + i32] = p;
            }
            // the other heads of a multi-output model: the same chain with their own w3 row
            for (uint32_t o = 1; o < n_out; ++o) {
#pragma unroll
                for (int mb = 0; mb < MBG; ++mb) {
                    float p = 0.0f;
#pragma unroll
                    for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 wv = *reinterpret_cast<const float4*>(w3s + o * H2 + (wave * NB2 + nb) * 32 + 8 * g + 4 * h);
                            p = __fmaf_rn(fmaxf(acc[mb][nb][4 * g + 0], 0.0f), wv.x, p);
                            p = __fmaf_rn(fmaxf(acc[mb][nb][4 * g + 1], 0.0f), wv.y, p);
                            p = __fmaf_rn(fmaxf(acc[mb][nb][4 * g + 2], 0.0f), wv.z, p);
                            p = __fmaf_rn(fmaxf(acc[mb][nb][4 * g + 3], 0.0f), wv.w, p);
                        }
                    hps[(o * 8 + wave * 2 + h) * kWsItems + (mbase + m0 + mb) * 32 + i32] = p;
                }
            }
        }
        RS_MARK(4)
        __syncthreads();                                   // partials visible; everyone is done with X and H1
        RS_MARK(5)

        // ---- scores: z = (((b3 + p0) + p1) + …) + p7, 16 items per wave; lane group g = lane / FPW takes heads g, g + 64 / FPW, …
        if (fin_item < cur.cnt) {
            for (uint32_t o = (uint32_t)lane / FPW; o < n_out; o += 64 / FPW) {
                float z = b3s[o];
#pragma unroll
                for (int s = 0; s < 8; ++s) z += hps[(o * 8 + s) * kWsItems + fin_item];
                a.out[(size_t)o * a.out_stride + cur.item0 + fin_item] = 1.0f / (1.0f + expf(-z));
            }
        }
        cur = nxt;
        nxt = nn;
        nn = uniform(d3);
        row_n1 = row_n2;
        RS_MARK(6)
    }
#ifdef PG_RS_PROFILE
    if (lane == 0 && blockIdx.x < 4) {
        uint64_t* o = (uint64_t*)(a.field_emb) + (blockIdx.x * 4 + (wave_id & 3)) * 8;
        for (int i = 0; i < 8; ++i) o[i] = ph[i];
    }
#endif
}

// WPC workgroups per CU: 128-128 needs 206 registers, so two of them share a CU (one's barriers and LDS phases run
// under the other's MFMAs); the wider shapes take 304 / 352 and run one
template <int H1, int H2, int WPC, int MBG, int MSPLIT, bool W1L>
static int launch_rs(pg_ctx* ctx, const MlpArgs& a) {
    const size_t lds = rs_lds_bytes<H1, H2, W1L>(a.n_out);
    int rc;
    if ((rc = ensure_dyn_lds(ctx, (const void*)dnn3_rs_kernel<H1, H2, WPC, MBG, MSPLIT, W1L>, lds))) return rc;
#ifdef PG_RS_PROFILE
    static uint64_t* dbg = nullptr;
    if (!dbg) hipMalloc(&dbg, 4 * 4 * 8 * 8);
    MlpArgs b = a;
    b.field_emb = reinterpret_cast<const float* const*>(dbg);
    dnn3_rs_kernel<H1, H2, WPC, MBG, MSPLIT, W1L><<<ctx->num_cus * WPC, 256 * MSPLIT, lds, ctx->stream>>>(b);
    uint64_t hcyc[128];
    hipMemcpy(hcyc, dbg, sizeof hcyc, hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls % 40 == 3)
        for (int wv = 0; wv < 8; ++wv) {
            fprintf(stderr, "rs<%d,%d> wg %d wave %d:", H1, H2, wv / 4, wv % 4);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %8llu", (unsigned long long)hcyc[wv * 8 + i]);
            fprintf(stderr, "\n");
        }
#else
    dnn3_rs_kernel<H1, H2, WPC, MBG, MSPLIT, W1L><<<ctx->num_cus * WPC, 256 * MSPLIT, lds, ctx->stream>>>(a);
#endif
    return PG_OK;
}

// ---------------------------------------------------------------------------------------------
// dnn3_ls_kernel<H1, H2>: DNN3 in bf16 for the LARGE hidden shape (1024-512), weights streamed.
// W1 (256 KB) + W2 (1 MB) fit neither the CU's registers nor its LDS, so they stream from L2 — once per 128-item tile
// (12.5 GB per 1.28 M items chip-wide; a 64-item tile would double that and be L2-bound).  The fp32 layer-2 accumulators
// of 128 items x 512 columns are 256 KB, half the register file, which is why mlp_kernel ran this shape with four waves
// of 256 accumulator registers, one per SIMD, every weight-fragment load exposed (wave wait 64 %, MFMA busy 22 %).
// Here EIGHT waves (two per SIMD, 256 registers each) share a tile: wave w owns output columns 64w..64w+63 of layer 2
// for all 128 items (128 accumulator registers).  Layer 1 runs in chunks of 64 hidden columns: wave w computes the
// 32 x 32 block (item block w & 3, column block w >> 2) of the chunk, relu → bf16 → a double-buffered 16-KB LDS tile;
// one barrier per chunk; layer 2 then adds the chunk's 64-deep partial product.  Weight fragments go global → registers
// (each is used by exactly one wave), single-buffered but re-requested as soon as their last MFMA has issued: layer 1's
// eight fragments of the next chunk right after this chunk's layer-1 MFMAs, layer 2's in two halves (k-steps 0-1, 2-3),
// so that every load has half a chunk of the other wave's MFMAs to arrive.
// ---------------------------------------------------------------------------------------------
#define LS_MFMA(acc, b, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(x))
#define LS_MFMA_READY(a0, a1) asm volatile("s_nop 3" : "+v"(a0), "+v"(a1))
#define LS_MFMA_DONE(a0, a1) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a0), "+v"(a1))
constexpr int kLsItems = 128;
// A 64-column bf16 operand tile has 128-B rows: two rows share the LDS's 256-B bank window, so the quad swizzle is keyed
// by (row >> 1) & 7 — rows of one parity then take eight different quads and 16 consecutive rows cover the window exactly
// (keyed by row & 7, as the shared store_h_quad does, rows r and r + 8 collide: every A-fragment read was 2-way
// conflicted, SQ_LDS_BANK_CONFLICT = 37 % of the LDS cycles of this kernel)
__device__ __forceinline__ void ls_store_h_quad(char* tile, int row, int col, float v0, float v1, float v2, float v3) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 lo = {v0, v1}, hi = {v2, v3};
    uint2 p;
    p.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));
    p.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2));
    *reinterpret_cast<uint2*>(tile + row * 128 + ((((col >> 3) ^ ((row >> 1) & 7))) << 4) + (col & 7) * 2) = p;
}
template <int H1, int H2>
constexpr size_t ls_lds_bytes(uint32_t n_out) {
    return (size_t)kLsItems * kDIN * 2 + 2 * (size_t)kLsItems * 64 * 2 + (size_t)(H1 + H2 + n_out * (H2 + 16 * kLsItems) + kMaxHeads) * 4;
}

template <int H1, int H2>
__global__ __launch_bounds__(512, 1) void dnn3_ls_kernel(MlpArgs a) {
    constexpr int M = kLsItems, CH = 64, NCH = H1 / CH, KS1 = kDIN / 16, KS2 = H1 / 16, KSC = CH / 16;
    static_assert(H2 == 512, "eight waves x 64 output columns");
    constexpr int XT_B = M * kDIN * 2, H1C_B = M * CH * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;
    char* const H1C = smem + XT_B;
    float* const c1s = reinterpret_cast<float*>(smem + XT_B + 2 * H1C_B);
    float* const b2s = c1s + H1;
    const uint32_t n_out = a.n_out;                        // heads on the shared trunk (1: the plain DNN3)
    float* const w3s = b2s + H2;                           // [n_out][H2]
    float* const hps = w3s + n_out * H2;                   // head partials [n_out][16 slots][128 items]
    float* const b3s = hps + n_out * 16 * kLsItems;        // [kMaxHeads]
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
    const char* const w1base = reinterpret_cast<const char*>(a.w1p);
    const char* const w2base = reinterpret_cast<const char*>(a.w2p) + (size_t)(wave * 2) * KS2 * 1024;
    const int mb1 = wave & 3, nb1 = wave >> 2;
    uint32_t c1_req = 0xffffffffu;

#ifdef PG_RS_PROFILE
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
#endif
    for (uint32_t tile = t_begin; tile < t_end; ++tile) {
        const uint32_t req = a.tile_req[tile], item0 = a.tile_item0[tile], cnt = a.tile_cnt[tile];
        uint32_t tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int i32 = tid_o & 31, h = (tid_o >> 5) & 1;
        const uint32_t lane_off = (tid_o & 63) * 16;       // unsigned: the loads then take the SGPR-base + 32-bit-offset form
        // ---- gather: 4 adjacent lanes per item, 64 contiguous bytes per instruction
        {
            const uint32_t g_item = tid_o >> 2, g_l = tid_o & 3;
            uint32_t row = a.cand_rows[item0 + (g_item < cnt ? g_item : cnt - 1)];
            row = row < a.tab_rows ? row : a.tab_rows - 1;
            const float4* src = reinterpret_cast<const float4*>(a.tab + (size_t)row * kDIN) + g_l;
            float4 xq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xq[j] = src[4 * j];
            if (req != c1_req) {
                c1_req = req;
                for (int i = tid_o; i < H1; i += 512) c1s[i] = a.c1[(size_t)req * a.c1_stride + i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) store_x_quad<1>(XT, g_item, 4 * j + g_l, xq[j]);
        }
        // ---- the first chunk's weight fragments
        bf16x8 w1f[KS1], w2f[2][KSC];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
            w1f[ks] = *reinterpret_cast<const bf16x8*>(w1base + (size_t)(nb1 * KS1 + ks) * 1024 + lane_off);
        f32x16 acc2[4][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(b2s + (wave * 2 + nb) * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    acc2[mb][nb][4 * g + 0] = bv.x;
                    acc2[mb][nb][4 * g + 1] = bv.y;
                    acc2[mb][nb][4 * g + 2] = bv.z;
                    acc2[mb][nb][4 * g + 3] = bv.w;
                }
            }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) LS_MFMA_READY(acc2[mb][0], acc2[mb][1]);
        RS_MARK(0)
        __syncthreads();
        RS_MARK(1)

        // layer 1 of chunk `cl` → H1 buffer `buf`; then its fragments are re-requested for chunk cl + 1
        auto layer1 = [&](int cl, int buf) {
            f32x16 acc1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 cv = *reinterpret_cast<const float4*>(c1s + cl * CH + nb1 * 32 + 8 * g + 4 * h);
                acc1[4 * g + 0] = cv.x;
                acc1[4 * g + 1] = cv.y;
                acc1[4 * g + 2] = cv.z;
                acc1[4 * g + 3] = cv.w;
            }
            const char* const xr = XT + (mb1 * 32 + i32) * 256;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(xr + (((ks * 2 + h) ^ (i32 & 15)) << 4));
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ks], af, acc1, 0, 0, 0);
            }
            const int cn = cl + 1 < NCH ? cl + 1 : cl;     // the last chunk re-requests its own fragments (unused)
            // (wave-uniform part of the address through readfirstlane: SGPR base + 32-bit lane offset, no 64-bit VGPR adds)
            const char* const w1n = w1base + (uint32_t)__builtin_amdgcn_readfirstlane((cn * 2 + nb1) * KS1 * 1024);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) w1f[ks] = *reinterpret_cast<const bf16x8*>(w1n + ks * 1024 + lane_off);
            char* const h1c = H1C + buf * H1C_B;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                ls_store_h_quad(h1c, mb1 * 32 + i32, nb1 * 32 + 8 * g + 4 * h, fmaxf(acc1[4 * g + 0], 0.0f),
                                fmaxf(acc1[4 * g + 1], 0.0f), fmaxf(acc1[4 * g + 2], 0.0f), fmaxf(acc1[4 * g + 3], 0.0f));
        };
        // layer 2 of chunk c: += H1 chunk (128 x 64) · W2[c*64 .. +64][64w .. +64]; its fragments are re-requested for
        // chunk cn in two halves.  The MFMAs are volatile asm (as in dnn3_ws_kernel): their order, and that of the LDS
        // reads written between them, is then the source order — A fragments three ahead (six MFMAs = 190 cycles) in four
        // rotating buffers.  Left to the scheduler this became read → s_waitcnt lgkmcnt(0) → two MFMAs, sixteen times.
        auto layer2 = [&](int c, int cn) {
            const char* const h1c = H1C + (c & 1) * H1C_B;
            const char* const w2n = w2base + (uint32_t)__builtin_amdgcn_readfirstlane(cn * KSC * 1024);
            bf16x8 af[4];
            auto frag = [&](int f) {                       // fragment f = (k-step f / 4, item block f % 4)
                return *reinterpret_cast<const bf16x8*>(h1c + ((f & 3) * 32 + i32) * (CH * 2) +
                                                        ((((f >> 2) * 2 + h) ^ ((i32 >> 1) & 7)) << 4));
            };
            af[0] = frag(0);
            af[1] = frag(1);
            af[2] = frag(2);
#pragma unroll
            for (int f = 0; f < 4 * KSC; ++f) {
                const int ks = f >> 2, mb = f & 3;
                if (f + 3 < 4 * KSC) af[(f + 3) & 3] = frag(f + 3);
                LS_MFMA(acc2[mb][0], w2f[0][ks], af[f & 3]);
                LS_MFMA(acc2[mb][1], w2f[1][ks], af[f & 3]);
                if ((ks == 1 || ks == 3) && mb == 3) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int k2 = ks - 1; k2 <= ks; ++k2)
                            w2f[nb][k2] = *reinterpret_cast<const bf16x8*>(w2n + (nb * KS2 + k2) * 1024 + lane_off);
                }
            }
        };
        auto first_w2f = [&]() {
#pragma unroll
            for (int kh = 0; kh < KSC; kh += 2)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int k2 = kh; k2 < kh + 2; ++k2)
                        w2f[nb][k2] = *reinterpret_cast<const bf16x8*>(w2base + (size_t)(nb * KS2 + k2) * 1024 + lane_off);
        };
        // Between two barriers a wave runs layer 1 of one chunk and layer 2 of another, and the two waves of a SIMD (w and
        // w + 4) are HALF AN INTERVAL APART: layer 1 is a chain of 8 dependent MFMAs between LDS reads, relu and stores
        // — latency, 1 300-2 000 cycles by itself — while layer 2 is 32 back-to-back MFMAs; in step, both waves idled the
        // matrix pipe together and then competed for it (PG_RS_PROFILE: 5 000 cycles per chunk for 2 600 of MFMA work).
        //   waves 0-3:  L1(0) b0 | L1(1) L2(0) b1 | L1(2) L2(1) b2 | …
        //   waves 4-7:  L1(0) b0 L2(0) | L1(1) b1 L2(1) | L1(2) b2 L2(2) | …
        // ONE code path (the same loop body, the barrier before or after its layer 2): L2(k) runs behind b_k, by which
        // every wave has written L1(k); L1(k + 2) overwrites the buffer L2(k) read only behind b_(k+1).  Two code paths
        // cost 261 spilled registers.  The first chunk's fragments are requested in the order the loop re-requests them —
        // the waitcnt pass merges the pending loads of the loop's entry and back edges, and where the orders differ, or a
        // path skips a group (hence waves 0-3 redo the last chunk's layer 1 into the idle buffer), it falls back to
        // counts that wait for loads issued moments ago: vmcnt(7) in front of the first MFMA of every chunk.
        const int ahead = wave >> 2;                       // waves 4-7 run layer 2 half an interval ahead
        layer1(0, 0);
        first_w2f();
        RS_MARK(2)
        __syncthreads();
        RS_MARK(3)
        if (ahead) layer2(0, 1);
        const int n_it = NCH - ahead;
#pragma unroll 1
        for (int c = 0; c < n_it; ++c) {
            const int c1 = c + 1 < NCH ? c + 1 : c;        // layer 1's chunk
            const int c2 = c + ahead;                      // layer 2's chunk
            layer1(c1, (c + 1) & 1);
            RS_MARK(2)
            if (ahead) __syncthreads();
            RS_MARK(3)
            layer2(c2, c2 + 1 < NCH ? c2 + 1 : c2);
            RS_MARK(4)
            if (!ahead) __syncthreads();
            RS_MARK(3)
        }
        if (ahead) __syncthreads();

        // ---- relu → dot head from the accumulators; 16 partials per item (wave, h), summed in slot order
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) LS_MFMA_DONE(acc2[mb][0], acc2[mb][1]);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            float p = 0.0f;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 wv = *reinterpret_cast<const float4*>(w3s + (wave * 2 + nb) * 32 + 8 * g + 4 * h);
                    p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 0], 0.0f), wv.x, p);
                    p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 1], 0.0f), wv.y, p);
                    p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 2], 0.0f), wv.z, p);
                    p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 3], 0.0f), wv.w, p);
                }
            hps[(wave * 2 + h) * M + mb * 32 + i32] = p;
        }
        for (uint32_t o = 1; o < n_out; ++o) {             // the other heads of a multi-output model
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float p = 0.0f;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 wv = *reinterpret_cast<const float4*>(w3s + o * H2 + (wave * 2 + nb) * 32 + 8 * g + 4 * h);
                        p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 0], 0.0f), wv.x, p);
                        p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 1], 0.0f), wv.y, p);
                        p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 2], 0.0f), wv.z, p);
                        p = __fmaf_rn(fmaxf(acc2[mb][nb][4 * g + 3], 0.0f), wv.w, p);
                    }
                hps[(o * 16 + wave * 2 + h) * M + mb * 32 + i32] = p;
            }
        }
        __syncthreads();
        // 512 threads, 128 items: thread group tid / 128 takes heads g, g + 4
        if ((tid_o & (M - 1)) < cnt) {
            for (uint32_t o = tid_o / M; o < n_out; o += 512 / M) {
                float z = b3s[o];
#pragma unroll
                for (int s = 0; s < 16; ++s) z += hps[(o * 16 + s) * M + (tid_o & (M - 1))];
                a.out[(size_t)o * a.out_stride + item0 + (tid_o & (M - 1))] = 1.0f / (1.0f + expf(-z));
            }
        }
        RS_MARK(5)
    }
#ifdef PG_RS_PROFILE
    if ((tid & 63) == 0 && blockIdx.x < 2) {
        uint64_t* o = (uint64_t*)(a.field_emb) + (blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = ph[i];
    }
#endif
}

template <int H1, int H2>
static int launch_ls(pg_ctx* ctx, const MlpArgs& a) {
    const size_t lds = ls_lds_bytes<H1, H2>(a.n_out);
    int rc;
    if ((rc = ensure_dyn_lds(ctx, (const void*)dnn3_ls_kernel<H1, H2>, lds))) return rc;
#ifdef PG_RS_PROFILE
    static uint64_t* dbg = nullptr;
    if (!dbg) hipMalloc(&dbg, 2 * 8 * 8 * 8);
    MlpArgs b = a;
    b.field_emb = reinterpret_cast<const float* const*>(dbg);
    dnn3_ls_kernel<H1, H2><<<ctx->num_cus, 512, lds, ctx->stream>>>(b);
    uint64_t hcyc[128];
    hipMemcpy(hcyc, dbg, sizeof hcyc, hipMemcpyDeviceToHost);
    static int calls = 0;
    if (++calls == 43)
        for (int wv = 0; wv < 16; ++wv) {
            fprintf(stderr, "ls<%d,%d> wg %d wave %d:", H1, H2, wv / 8, wv % 8);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %8llu", (unsigned long long)hcyc[wv * 8 + i]);
            fprintf(stderr, "\n");
        }
#else
    dnn3_ls_kernel<H1, H2><<<ctx->num_cus, 512, lds, ctx->stream>>>(a);
#endif
    return PG_OK;
}

bool dnn3_ls_shape(uint32_t h1, uint32_t h2) { return h1 == 1024 && h2 == 512; }
int launch_dnn3_ls(pg_ctx* ctx, uint32_t h1, uint32_t h2, const MlpArgs& a) {
    if (h1 == 1024 && h2 == 512) return launch_ls<1024, 512>(ctx, a);
    set_error("rank: no streamed-weights kernel for hidden widths %u-%u", h1, h2);
    return PG_ERR_UNSUPPORTED;
}

bool dnn3_rs_shape(uint32_t h1, uint32_t h2) {
    return (h1 == 128 && h2 == 128) || (h1 == 256 && h2 == 128) || (h1 == 256 && h2 == 256);
}

int launch_dnn3_rs(pg_ctx* ctx, uint32_t h1, uint32_t h2, const MlpArgs& a) {
    if (h1 == 128 && h2 == 128) return launch_rs<128, 128, 2, 2, 1, false>(ctx, a);
    if (h1 == 256 && h2 == 128) return launch_rs<256, 128, 2, 1, 1, false>(ctx, a);
    if (h1 == 256 && h2 == 256) return launch_rs<256, 256, 1, 1, 2, true>(ctx, a);
    set_error("rank: no register-stationary kernel for hidden widths %u-%u", h1, h2);
    return PG_ERR_UNSUPPORTED;
}

}  // namespace pg
