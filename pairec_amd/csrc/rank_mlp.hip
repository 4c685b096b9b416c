// rank_mlp.hip — the rank stage's model predict on MFMA.
//
// Replaces the DNN / FM forward that the reference ships to a remote model server:
// EasModel.Run (algorithm/eas/model.go:197-222), TFservingModel.Run
// (algorithm/tfserving/model.go:30-55), called once per batch of 100 items from
// RankService.Rank (service/rank/rank_service.go:163-166,264-289).  Here one launch scores every
// candidate of every request in the call; candidate features are gathered straight from HBM.
//
// One fused kernel serves both model families:
//     z1 = c1[req] + x · W1        h1 = P(relu(z1))           x = 128 gathered fp32 values / item
//     z2 = b2 + h1 · W2            h2 = act2(z2)   (kept fp32)
//     z3 = bias3[item] + <h2, w3[req]>  (two half chains)      score = 1/(1+expf(-z3))
//   DNN3      : x = table row, c1 = b1 + user·W1[user half] (request constant), w3 shared, bias3=b3
//   two-tower : x = concat of 8 item-field embeddings, c1 = ib1, w3 = user-tower output, bias3=y_fm
// PG_PREC_BF16: operands bf16, fp32 accumulate on v_mfma_f32_32x32x16_bf16.
// PG_PREC_F32 : v_mfma_f32_32x32x2_f32, which is a k-ordered fmaf chain → bit-reproducible.
// PG_PREC_BF16X3: the fp32 specification on the bf16 matrix pipe — activations and weights as hi + lo bf16 pairs,
//               three products (lo·hi, hi·lo, hi·hi) into the fp32 accumulator; scores within 1e-5 of the fp32 path
//               (observed ~1e-6), no operand rounding anywhere else.  Model outputs are fp32 widened to f64 in the
//               reference (algorithm/eas/easyrec_response.go:479-483, eas/tf_response.go:55-59).
//
// Tiling: workgroup = 4 waves = 128 items; layer 1 is produced in chunks of 128 hidden columns
// that go through LDS (as the A operand of layer 2) and are consumed immediately, so h1 never
// leaves the CU; weights are pre-packed in MFMA-fragment order so each B fragment is one
// coalesced 1 KiB load shared by all of a wave's row blocks.
#include "rank_mlp.hpp"

#include <algorithm>
#include <cmath>

namespace pg {

// C[rows of this wave][n-blocks] += A(tile in LDS)[rows][K] · B(pre-packed fragments)
// frag(nb, step) returns the byte offset of the 1-KiB fragment for n-block nb and k-group `step`
// (bf16: 16 k per group, f32: 8 k per group).
// TR = true swaps the MFMA operands: the accumulator block then holds C^T — lane ↔ item row, register r ↔
// column (r&3) + 8(r>>2) + 4h of the n-block — so a lane owns 4 consecutive columns of one item, which
// packs into one LDS store (the products and the k order are unchanged, so the bits are too).
// PREC 2 (split bf16): `tile` is the hi tile, the lo tile lies tile_lo bytes on; wpk_lo = the lo fragments.
template <int PREC, int MB, int NB, int K, bool TR, typename FragOff>
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[MB][NB], const char* tile, int mrow0,
                                          const char* wpk, FragOff frag, int lane, int tile_lo = 0,
                                          const char* wpk_lo = nullptr) {
    constexpr int ES = PREC ? 2 : 4;
    constexpr int ROWB = K * ES;
    constexpr int SW = (ROWB / 16 < 16 ? ROWB / 16 : 16) - 1;
    const int i32 = lane & 31, h = lane >> 5;
    if constexpr (PREC == 2) {
        static_assert(TR, "split bf16: transposed accumulators only");
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
            bf16x8 bh[NB], bl[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                bh[nb] = *reinterpret_cast<const bf16x8*>(wpk + frag(nb, ks) + lane * 16);
                bl[nb] = *reinterpret_cast<const bf16x8*>(wpk_lo + frag(nb, ks) + lane * 16);
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int row = mrow0 + mb * 32 + i32;
                const char* const ap = tile + row * ROWB + ((((ks * 2 + h) ^ (row & SW))) << 4);
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + tile_lo);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {      // the two corrections first, the leading product last
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[nb], ah, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nb], al, acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[nb], ah, acc[mb][nb], 0, 0, 0);
                }
            }
        }
    } else if constexpr (PREC == 1) {
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
            bf16x8 bf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bf[nb] = *reinterpret_cast<const bf16x8*>(wpk + frag(nb, ks) + lane * 16);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int row = mrow0 + mb * 32 + i32;
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(
                    tile + row * ROWB + ((((ks * 2 + h) ^ (row & SW))) << 4));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = TR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[nb], af, acc[mb][nb], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[nb], acc[mb][nb], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int g8 = 0; g8 < K / 8; ++g8) {       // groups of 8 k (4 MFMA steps each)
            f32x4 bf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bf[nb] = *reinterpret_cast<const f32x4*>(wpk + frag(nb, g8) + lane * 16);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int row = mrow0 + mb * 32 + i32;
#pragma unroll
                for (int qd = 0; qd < 2; ++qd) {
                    const f32x4 aq = *reinterpret_cast<const f32x4*>(
                        tile + row * ROWB + ((((g8 * 2 + qd) ^ (row & SW))) << 4));
                    const float a0 = h ? aq.y : aq.x;
                    const float a1 = h ? aq.w : aq.z;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (TR) {
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[nb][2 * qd], a0, acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[nb][2 * qd + 1], a1, acc[mb][nb], 0, 0, 0);
                        } else {
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bf[nb][2 * qd], acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bf[nb][2 * qd + 1], acc[mb][nb], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
}

// bf16 path with the B fragments of a whole GEMM loaded ahead of time (so that their L2 latency hides
// behind the previous GEMM / the H1 store and the barrier instead of stalling every other MFMA)
template <int NB, int KS, typename FragOff>
__device__ __forceinline__ void load_bfrags(bf16x8 (&bfr)[KS][NB], const char* wpk, FragOff frag, int lane) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            bfr[ks][nb] = *reinterpret_cast<const bf16x8*>(wpk + frag(nb, ks) + lane * 16);
}
template <int MB, int NB, int K>
__device__ __forceinline__ void gemm_tile_pre(f32x16 (&acc)[MB][NB], const char* tile, int mrow0,
                                              const bf16x8 (&bfr)[K / 16][NB], int lane) {
    constexpr int ROWB = K * 2;
    constexpr int SW = (ROWB / 16 < 16 ? ROWB / 16 : 16) - 1;
    const int i32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int row = mrow0 + mb * 32 + i32;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(tile + row * ROWB + ((((ks * 2 + h) ^ (row & SW))) << 4));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)     // operands swapped: transposed accumulators (see gemm_tile)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][nb], af, acc[mb][nb], 0, 0, 0);
        }
}

// LDS of one workgroup: X tile + H1 chunk tile, aliased after the GEMMs by the fp32 H2 tile of one
// epilogue pass (H2 / EP columns, padded rows); then w3 and the per-item head bias.
constexpr size_t mlp_lds_bytes(int prec, int h2, int ch, int ep, int bm = kBM) {
    const size_t es = prec == 1 ? 2 : 4;               // (split bf16: a hi and a lo tile of 2 B per element)
    const size_t nh1 = (prec == 1 && (size_t)bm * (kDIN + 2 * ch) * es <= 72 * 1024) ? 2 : 1;   // as NH1 in mlp_kernel
    const size_t tiles = (size_t)bm * (kDIN + nh1 * ch) * es;
    const size_t h2t = (size_t)bm * (h2 / ep + 4) * 4;
    return (tiles > h2t ? tiles : h2t) + (size_t)h2 * 4 + (size_t)bm * 4;
}

// MODEL 1 = DNN3, 2 = two-tower item side.  WM x WN = wave grid over (items, hidden columns); CH = layer-1
// chunk width (hidden columns produced, pushed through LDS and consumed by layer 2 at a time); EP = passes of
// the head epilogue (the fp32 H2 tile goes through LDS H2/EP columns at a time — EP = 2 halves the LDS
// footprint so that two workgroups share a CU); OCC = workgroups per CU the register budget is set for.
// FK: two-tower only — width of a field embedding; the item side has kDIN / FK fields (8 x 16, 4 x 32, 16 x 8).
// BM: items per workgroup tile (128; 64 where more, smaller workgroups per CU hide the gathers' latency better).
template <int PREC, int H1, int H2, bool ACT2, int WM, int WN, int MODEL, int CH, int EP, int OCC, int FK = 16, int BM = kBM>
__global__ __launch_bounds__(256, OCC) void mlp_kernel(MlpArgs a) {
    constexpr int MB = BM / 32 / WM;
    constexpr int NP = BM / 8;                         // gather passes: 8 items (32 lanes x 16 B each) per pass
    constexpr int L1NB = CH / 32 / WN;
    constexpr int L2NB = H2 / 32 / WN;
    constexpr int ES = PREC ? 2 : 4;
    constexpr int PL = PREC == 2 ? 2 : 1;              // operand tiles per matrix (split bf16: hi, then lo)
    constexpr int X_LO = BM * kDIN * ES;               // offset of the lo tile (PREC 2)
    constexpr int H_LO = BM * CH * ES;
    constexpr int TILE_B = BM * kDIN * ES * PL;
    constexpr int H1_B = BM * CH * ES * PL;
    constexpr int NCHUNK = H1 / CH;
    constexpr int KG1 = PREC ? kDIN / 16 : kDIN / 8;   // k-groups (fragments) of the 128-deep layer-1 GEMM
    constexpr int KGC = PREC ? CH / 16 : CH / 8;       // k-groups of one CH-deep layer-2 partial GEMM
    constexpr int HH = H2 / EP;                        // head columns per epilogue pass
    // H1 chunk tiles: double-buffered where two workgroups still fit a CU's 160 KB
    constexpr int NH1 = (PREC == 1 && TILE_B + 2 * H1_B <= 72 * 1024) ? 2 : 1;
    constexpr size_t REGION = (size_t)(TILE_B + NH1 * H1_B) > (size_t)BM * (HH + 4) * 4
                                  ? (size_t)(TILE_B + NH1 * H1_B) : (size_t)BM * (HH + 4) * 4;
    static_assert(L1NB >= 1 && L2NB >= 1 && MB >= 1, "bad wave layout");
    static_assert(EP == 1 || (EP == 2 && HH % 32 == 0), "epilogue passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;
    char* const H1T0 = smem + TILE_B;
    float* const H2T = reinterpret_cast<float*>(smem);      // aliases XT/H1T after the GEMMs
    float* const w3s = reinterpret_cast<float*>(smem + REGION);
    float* const b3s = w3s + H2;

    const uint32_t tile = blockIdx.x;
    if (tile >= *a.n_tiles) return;
    const uint32_t req = a.tile_req[tile];
    const uint32_t item0 = a.tile_item0[tile];
    const uint32_t cnt = a.tile_cnt[tile];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int i32 = lane & 31, h = lane >> 5;

    // fragment offsets: W1 n-block (chunk, wave, nb), k-group step; W2 n-block (wave, nb), k-group chunk*KGC + step
    constexpr int KG2 = NCHUNK * KGC;                  // k-groups over the full H1 depth
    auto frag1 = [&](int chunk) {
        const int nbg0 = chunk * (CH / 32) + wn * L1NB;
        return [=](int nb, int step) { return (size_t)((nbg0 + nb) * KG1 + step) * 1024; };
    };
    auto frag2 = [&](int chunk) {
        const int nbg0 = wn * L2NB;
        return [=](int nb, int step) { return (size_t)((nbg0 + nb) * KG2 + chunk * KGC + step) * 1024; };
    };
    constexpr bool PRE = PREC == 1 && OCC >= 2;        // preloaded B fragments (bf16, several workgroups per CU)
    bf16x8 b1f[PRE ? KG1 : 1][PRE ? L1NB : 1];
    bf16x8 b2f[PRE ? KGC : 1][PRE ? L2NB : 1];
#ifdef PG_MLP_PROFILE
    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
#define MLP_MARK(i) { const uint64_t tn = __builtin_readcyclecounter(); prof[i] += tn - tp; tp = tn; }
#else
#define MLP_MARK(i)
#endif

    // ---------------- gather prologue: 32 lanes x 16 B per item, 8 items per pass ------------
    {
        const int c = tid & 31;
        for (int i = tid; i < H2; i += 256) w3s[i] = a.w3[(size_t)req * a.w3_stride + i];
        if constexpr (MODEL == 1) {
            if (tid < BM) b3s[tid] = a.b3;
            float4 v[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const uint32_t r = p * 8 + (tid >> 5);
                const uint32_t idx = item0 + (r < cnt ? r : cnt - 1);
                uint32_t row = a.cand_rows[idx];
                row = row < a.tab_rows ? row : a.tab_rows - 1;
                v[p] = (uint32_t)(4 * c) < a.tab_dim ? *reinterpret_cast<const float4*>(a.tab + (size_t)row * a.tab_dim + 4 * c)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) store_x_quad<PREC>(XT, p * 8 + (tid >> 5), c, v[p], X_LO);
        } else if constexpr (MODEL == 3) {
            // two-tower from MATERIALISED item records (pg_fm2t_item_rows_build): the item's field embeddings and linear
            // weights are one contiguous 544-B record found from the candidate row alone — no ids → rows second hop, five
            // 128-B lines per item instead of eight (a 64-B embedding row costs a whole line) plus the id line.  All 256
            // threads gather, TWO ADJACENT LANES PER RECORD: lane kh of the pair takes the 16-B quads 2j + kh of every
            // field, so a load instruction covers 32 contiguous bytes per record.  (One lane per record — 64 different
            // records per instruction — reaches 1.5 TB/s on a 12.8 GB catalogue, two lanes per record 5.8 TB/s:
            // scripts/micro/gather_rec.hip.)  Each lane accumulates its columns of s_k / q_k in the specification's order
            // (user prefix first, fields ascending) and reduces the levels of the balanced tree that stay inside its
            // quads; the level that pairs quad 2m with quad 2m + 1 crosses the lane pair (one exchange per m), the rest
            // is lane 0's — the same operations in the same order as the per-field path: bit-identical.
            constexpr int NF = kDIN / FK;          // item fields
            constexpr int QPF = FK / 4;            // 16-B quads per field
            constexpr int QPH = QPF / 2;           // quads of a field per lane
            constexpr int HK = QPH * 4;            // columns of a field per lane
            static_assert(BM == 128 && QPF % 2 == 0, "MODEL 3: 128-item tiles, field width a multiple of 8");
            (void)c;
            const uint32_t r = (uint32_t)tid >> 1;
            const int kh = tid & 1;
            const uint32_t idx = item0 + (r < cnt ? r : cnt - 1);
            uint32_t row = a.cand_rows[idx];
            row = row < a.irow_count ? row : a.irow_count;
            const float4* rec = reinterpret_cast<const float4*>(a.irows + (size_t)row * kItemRowFloats);
            const float* fu = a.fm_user + (size_t)req * kFmUserStride;
            float4 v[NF][QPH];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int j = 0; j < QPH; ++j) v[f][j] = rec[f * QPF + 2 * j + kh];
            float linv[NF];
            if (kh == 0) {
#pragma unroll
                for (int j = 0; j < (NF + 3) / 4; ++j) {
                    const float4 l4 = rec[kDIN / 4 + j];
                    linv[4 * j + 0] = l4.x;
                    if (4 * j + 1 < NF) linv[4 * j + 1] = l4.y;
                    if (4 * j + 2 < NF) linv[4 * j + 2] = l4.z;
                    if (4 * j + 3 < NF) linv[4 * j + 3] = l4.w;
                }
            }
            float s_[HK], q_[HK];                  // local column 4j + i is column 4 (2j + kh) + i of the field
#pragma unroll
            for (int j = 0; j < QPH; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s_[4 * j + i] = fu[1 + 4 * (2 * j + kh) + i];
                    q_[4 * j + i] = fu[1 + kFmMaxK + 4 * (2 * j + kh) + i];
                }
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int j = 0; j < QPH; ++j) {
                    const float4 x = v[f][j];
                    s_[4 * j + 0] = s_[4 * j + 0] + x.x; q_[4 * j + 0] = __fmaf_rn(x.x, x.x, q_[4 * j + 0]);
                    s_[4 * j + 1] = s_[4 * j + 1] + x.y; q_[4 * j + 1] = __fmaf_rn(x.y, x.y, q_[4 * j + 1]);
                    s_[4 * j + 2] = s_[4 * j + 2] + x.z; q_[4 * j + 2] = __fmaf_rn(x.z, x.z, q_[4 * j + 2]);
                    s_[4 * j + 3] = s_[4 * j + 3] + x.w; q_[4 * j + 3] = __fmaf_rn(x.w, x.w, q_[4 * j + 3]);
                    store_x_quad<PREC>(XT, (int)r, f * QPF + 2 * j + kh, x, X_LO);
                }
#pragma unroll
            for (int k = 0; k < HK; ++k) s_[k] = __fmaf_rn(s_[k], s_[k], -q_[k]);
            // balanced pairwise tree over the field's FK columns: levels 1 and 2 stay inside a quad ...
#pragma unroll
            for (int j = 0; j < QPH; ++j) {
                s_[4 * j + 0] = s_[4 * j + 0] + s_[4 * j + 1];
                s_[4 * j + 2] = s_[4 * j + 2] + s_[4 * j + 3];
                s_[4 * j + 0] = s_[4 * j + 0] + s_[4 * j + 2];
            }
            // ... level 4 adds quad 2m + 1 (the odd lane's) to quad 2m (the even lane's) ...
#pragma unroll
            for (int j = 0; j < QPH; ++j) s_[4 * j] = s_[4 * j] + __shfl_xor(s_[4 * j], 1);
            // ... and the levels above pair the even lane's quads (column 8m ↔ local 4m)
#pragma unroll
            for (int off = 1; off < QPH; off <<= 1)
#pragma unroll
                for (int j = 0; j < QPH; j += 2 * off) s_[4 * j] = s_[4 * j] + s_[4 * (j + off)];
            if (kh == 0) {
                float lin = fu[0];
#pragma unroll
                for (int f = 0; f < NF; ++f) lin = lin + linv[f];
                b3s[r] = lin + 0.5f * s_[0];
            }
        } else {
            // two-tower: ONE THREAD PER ITEM.  The thread reads its item's field ids, then walks the fields in order:
            // 64 B (FK = 16) of each field's embedding row as 16-B loads, FM sums s_k / q_k accumulated in registers in
            // the specification's order (user prefix first, fields ascending — no cross-lane traffic at all), the row
            // stored into the X tile as it goes.  The loads of all fields are independent (they only need the ids), so
            // the compiler keeps them in flight together.  Versus the previous layout — 32 lanes per item, 16 passes,
            // field sums through an LDS staging area — this is ~6x fewer VALU instructions per item: profiling showed
            // 16 VALU per MFMA and waves issue-stalled 45 % of the time (profiles/r2_cfg4_*).
            constexpr int NF = kDIN / FK;          // item fields
            constexpr int QPF = FK / 4;            // 16-B quads per field embedding
            (void)c;
            if (tid < BM) {
                const uint32_t r = (uint32_t)tid;
                const uint32_t idx = item0 + (r < cnt ? r : cnt - 1);
                const float* fu = a.fm_user + (size_t)req * kFmUserStride;
                int32_t ids[NF];
                {
                    const int4* ip = reinterpret_cast<const int4*>(a.item_field_ids + (size_t)idx * NF);
#pragma unroll
                    for (int j = 0; j < NF / 4; ++j) {
                        const int4 t4 = ip[j];
                        ids[4 * j + 0] = t4.x; ids[4 * j + 1] = t4.y; ids[4 * j + 2] = t4.z; ids[4 * j + 3] = t4.w;
                    }
                }
                float4 v[NF][QPF];
                float linv[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    int32_t id = ids[f];
                    id = id < 0 ? 0 : (id >= (int32_t)a.vocab ? (int32_t)a.vocab - 1 : id);
                    const float4* e4 = reinterpret_cast<const float4*>(a.field_emb[a.n_user_fields + f] + (size_t)id * FK);
#pragma unroll
                    for (int j = 0; j < QPF; ++j) v[f][j] = e4[j];
                    linv[f] = a.field_lin[a.n_user_fields + f][id];
                }
                float s_[FK], q_[FK];
#pragma unroll
                for (int k = 0; k < FK; ++k) {
                    s_[k] = fu[1 + k];
                    q_[k] = fu[1 + kFmMaxK + k];
                }
                float lin = fu[0];
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    lin = lin + linv[f];
#pragma unroll
                    for (int j = 0; j < QPF; ++j) {
                        const float4 x = v[f][j];
                        s_[4 * j + 0] = s_[4 * j + 0] + x.x; q_[4 * j + 0] = __fmaf_rn(x.x, x.x, q_[4 * j + 0]);
                        s_[4 * j + 1] = s_[4 * j + 1] + x.y; q_[4 * j + 1] = __fmaf_rn(x.y, x.y, q_[4 * j + 1]);
                        s_[4 * j + 2] = s_[4 * j + 2] + x.z; q_[4 * j + 2] = __fmaf_rn(x.z, x.z, q_[4 * j + 2]);
                        s_[4 * j + 3] = s_[4 * j + 3] + x.w; q_[4 * j + 3] = __fmaf_rn(x.w, x.w, q_[4 * j + 3]);
                        store_x_quad<PREC>(XT, (int)r, f * QPF + j, x, X_LO);
                    }
                }
#pragma unroll
                for (int k = 0; k < FK; ++k) s_[k] = __fmaf_rn(s_[k], s_[k], -q_[k]);
#pragma unroll
                for (int off = 1; off < FK; off <<= 1)          // balanced pairwise tree over k
#pragma unroll
                    for (int k = 0; k < FK; k += 2 * off) s_[k] = s_[k] + s_[k + off];
                b3s[r] = lin + 0.5f * s_[0];
            }
        }
    }
    MLP_MARK(0)
    __syncthreads();
    MLP_MARK(1)

    const int mrow0 = wm * MB * 32;
    f32x16 acc2[MB][L2NB];
#pragma unroll
    for (int nb = 0; nb < L2NB; ++nb) {          // transposed like acc1: register r ↔ column (r&3)+8(r>>2)+4h
        const float* bb = a.b2 + (wn * L2NB + nb) * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4*>(bb + 8 * g);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                acc2[mb][nb][4 * g + 0] = bv.x;
                acc2[mb][nb][4 * g + 1] = bv.y;
                acc2[mb][nb][4 * g + 2] = bv.z;
                acc2[mb][nb][4 * g + 3] = bv.w;
            }
        }
    }

    if constexpr (PRE) load_bfrags<L1NB, KG1>(b1f, reinterpret_cast<const char*>(a.w1p), frag1(0), lane);

#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        // with two H1 tiles the barrier after the layer-2 GEMM is not needed: the next chunk writes the
        // other tile, and the barrier before ITS layer-2 GEMM orders everything two chunks apart
        char* const H1T = H1T0 + (NH1 == 2 ? (chunk & 1) * H1_B : 0);
        // ---- layer 1, columns [chunk*128, +128): wave owns L1NB n-blocks
        // (transposed accumulators: lane ↔ item mb*32 + i32, register r ↔ hidden column
        //  (r&3) + 8(r>>2) + 4h of the n-block)
        f32x16 acc1[MB][L1NB];
#pragma unroll
        for (int nb = 0; nb < L1NB; ++nb) {
            const float* cb = a.c1 + (size_t)req * a.c1_stride + chunk * CH + (wn * L1NB + nb) * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 cv = *reinterpret_cast<const float4*>(cb + 8 * g);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    acc1[mb][nb][4 * g + 0] = cv.x;
                    acc1[mb][nb][4 * g + 1] = cv.y;
                    acc1[mb][nb][4 * g + 2] = cv.z;
                    acc1[mb][nb][4 * g + 3] = cv.w;
                }
            }
        }
        if constexpr (PRE) {
            gemm_tile_pre<MB, L1NB, kDIN>(acc1, XT, mrow0, b1f, lane);
            load_bfrags<L2NB, KGC>(b2f, reinterpret_cast<const char*>(a.w2p), frag2(chunk), lane);
        } else {
            gemm_tile<PREC, MB, L1NB, kDIN, true>(acc1, XT, mrow0, reinterpret_cast<const char*>(a.w1p),
                                                  frag1(chunk), lane, X_LO, reinterpret_cast<const char*>(a.w1p_lo));
        }
        MLP_MARK(2)
        // relu → P() → H1 chunk tile (A operand of layer 2): one packed store per 4 columns
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < L1NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = mrow0 + mb * 32 + i32;
                    const int col = (wn * L1NB + nb) * 32 + 8 * g + 4 * h;
                    const float v0 = acc1[mb][nb][4 * g + 0], v1 = acc1[mb][nb][4 * g + 1];
                    const float v2 = acc1[mb][nb][4 * g + 2], v3 = acc1[mb][nb][4 * g + 3];
                    store_h_quad<PREC, CH>(H1T, row, col, v0 > 0.0f ? v0 : 0.0f, v1 > 0.0f ? v1 : 0.0f,
                                           v2 > 0.0f ? v2 : 0.0f, v3 > 0.0f ? v3 : 0.0f, H_LO);
                }
        MLP_MARK(3)
        __syncthreads();
        MLP_MARK(4)
        // ---- layer 2 partial: acc2 += H1chunk · W2[chunk*CH .. +CH, :]
        if constexpr (PRE) {
            gemm_tile_pre<MB, L2NB, CH>(acc2, H1T, mrow0, b2f, lane);
            if (chunk + 1 < NCHUNK)      // next chunk's layer-1 fragments: in flight across the barrier
                load_bfrags<L1NB, KG1>(b1f, reinterpret_cast<const char*>(a.w1p), frag1(chunk + 1), lane);
        } else {
            gemm_tile<PREC, MB, L2NB, CH, true>(acc2, H1T, mrow0, reinterpret_cast<const char*>(a.w2p),
                                                frag2(chunk), lane, H_LO, reinterpret_cast<const char*>(a.w2p_lo));
        }
        MLP_MARK(5)
        if (NH1 == 1 || chunk + 1 == NCHUNK) __syncthreads();
        MLP_MARK(6)
    }

    // ---- layer-2 activation → H2 tile (fp32; rows padded by one 16-B quad: an odd number of quads per
    // row keeps both the float4 stores and the per-thread float4 row walks conflict-free) → dot head: two
    // half chains over the H2 columns, z = (b3 + Σ_{m < H2/2}) + Σ_{m >= H2/2}, each
    // sequential in m.
    constexpr int HS = HH + 4;                            // H2 tile row stride in floats
    auto relu2 = [](float v) { return ACT2 ? (v > 0.0f ? v : 0.0f) : v; };
    if constexpr (EP == 1) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < L2NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = mrow0 + mb * 32 + i32;
                    const int col = (wn * L2NB + nb) * 32 + 8 * g + 4 * h;
                    *reinterpret_cast<float4*>(H2T + row * HS + col) =
                        make_float4(relu2(acc2[mb][nb][4 * g + 0]), relu2(acc2[mb][nb][4 * g + 1]),
                                    relu2(acc2[mb][nb][4 * g + 2]), relu2(acc2[mb][nb][4 * g + 3]));
                }
        __syncthreads();
        const int row = (tid >> 1) < BM ? (tid >> 1) : BM - 1, half = tid & 1;      // (BM < 128: the surplus threads idle along)
        const float4* hr = reinterpret_cast<const float4*>(H2T + row * HS + half * (H2 / 2));
        const float4* wr = reinterpret_cast<const float4*>(w3s + half * (H2 / 2));
        float p = half ? 0.0f : b3s[row];
#pragma unroll 4
        for (int m = 0; m < H2 / 8; ++m) {
            const float4 x = hr[m], y = wr[m];
            p = __fmaf_rn(x.x, y.x, p);
            p = __fmaf_rn(x.y, y.y, p);
            p = __fmaf_rn(x.z, y.z, p);
            p = __fmaf_rn(x.w, y.w, p);
        }
        const float o = __shfl_xor(p, 1);
        const float z = half ? (o + p) : (p + o);
        if (half == 0 && (tid >> 1) < BM && (uint32_t)row < cnt) a.out[item0 + row] = 1.0f / (1.0f + expf(-z));
        if constexpr (MODEL == 1) {
            // the other heads of a multi-output model (easyrec_response.go:35-70): the same two half chains over the same
            // H2 tile, w3 row `hd` straight from global memory (every thread reads the same addresses)
            for (uint32_t hd = 1; hd < a.n_out; ++hd) {
                const float4* wg = reinterpret_cast<const float4*>(a.w3 + (size_t)hd * H2 + half * (H2 / 2));
                float ph_ = half ? 0.0f : a.b3v[hd];
#pragma unroll 4
                for (int m = 0; m < H2 / 8; ++m) {
                    const float4 x = hr[m], y = wg[m];
                    ph_ = __fmaf_rn(x.x, y.x, ph_);
                    ph_ = __fmaf_rn(x.y, y.y, ph_);
                    ph_ = __fmaf_rn(x.z, y.z, ph_);
                    ph_ = __fmaf_rn(x.w, y.w, ph_);
                }
                const float oh = __shfl_xor(ph_, 1);
                const float zh = half ? (oh + ph_) : (ph_ + oh);
                if (half == 0 && (tid >> 1) < BM && (uint32_t)row < cnt)
                    a.out[(size_t)hd * a.out_stride + item0 + row] = 1.0f / (1.0f + expf(-zh));
            }
        }
    } else {
        // pass e carries columns [e*HH, (e+1)*HH) = half chain e; thread `row` (tid < 128) runs both
        float ph[2] = {0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < EP; ++e) {
            if (e) __syncthreads();                       // the previous pass's readers are done
#pragma unroll
            for (int nb = 0; nb < L2NB; ++nb) {
                const int col0 = (wn * L2NB + nb) * 32;
                if (col0 / HH != e) continue;             // wave-uniform
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = mrow0 + mb * 32 + i32;
                        *reinterpret_cast<float4*>(H2T + row * HS + (col0 - e * HH) + 8 * g + 4 * h) =
                            make_float4(relu2(acc2[mb][nb][4 * g + 0]), relu2(acc2[mb][nb][4 * g + 1]),
                                        relu2(acc2[mb][nb][4 * g + 2]), relu2(acc2[mb][nb][4 * g + 3]));
                    }
            }
            __syncthreads();
            if (tid < BM) {
                const float4* hr = reinterpret_cast<const float4*>(H2T + tid * HS);
                const float4* wr = reinterpret_cast<const float4*>(w3s + e * HH);
                float p = e ? 0.0f : b3s[tid];
#pragma unroll 4
                for (int m = 0; m < HH / 4; ++m) {
                    const float4 x = hr[m], y = wr[m];
                    p = __fmaf_rn(x.x, y.x, p);
                    p = __fmaf_rn(x.y, y.y, p);
                    p = __fmaf_rn(x.z, y.z, p);
                    p = __fmaf_rn(x.w, y.w, p);
                }
                ph[e] = p;
                if constexpr (MODEL == 1) {
                    // the other heads: half chain e of head `hd`; pass 0 parks it in the output slot, pass 1 finishes it
                    for (uint32_t hd = 1; hd < a.n_out; ++hd) {
                        const float4* wg = reinterpret_cast<const float4*>(a.w3 + (size_t)hd * H2 + e * HH);
                        float q = e ? 0.0f : a.b3v[hd];
#pragma unroll 4
                        for (int m = 0; m < HH / 4; ++m) {
                            const float4 x = hr[m], y = wg[m];
                            q = __fmaf_rn(x.x, y.x, q);
                            q = __fmaf_rn(x.y, y.y, q);
                            q = __fmaf_rn(x.z, y.z, q);
                            q = __fmaf_rn(x.w, y.w, q);
                        }
                        if ((uint32_t)tid < cnt) {
                            float* dst = a.out + (size_t)hd * a.out_stride + item0 + tid;
                            *dst = e == 0 ? q : 1.0f / (1.0f + expf(-(*dst + q)));
                        }
                    }
                }
            }
        }
        if (tid < BM && (uint32_t)tid < cnt) a.out[item0 + tid] = 1.0f / (1.0f + expf(-(ph[0] + ph[1])));
    }
#ifdef PG_MLP_PROFILE
    if constexpr (MODEL == 3 && PREC == 1) {           // (field_emb is unused by this instance: the launcher points it at a buffer)
        MLP_MARK(7)
        if (lane == 0 && blockIdx.x >= 5000 && blockIdx.x < 5016) {
            uint64_t* o = (uint64_t*)(a.field_emb) + ((blockIdx.x - 5000) * 4 + wave) * 8;
            for (int i = 0; i < 8; ++i) o[i] = prof[i];
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// request → tile table.  One block of 1024 threads; requests <= 65535 per call.
//   pass 1: tiles per request → exclusive prefix req_tile0 (chunked scan)
//   pass 2: one thread per TILE (binary search of its request in the prefix), so the three tables are written
//           with coalesced stores (one thread per request wrote them with a stride of a request's tile count:
//           60 us for 256 requests x 79 tiles — as long as a tenth of the MLP it feeds)
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kTilesLdsReqs = 8192;     // prefix entries searched in LDS; beyond that, in global memory

__global__ __launch_bounds__(1024) void build_tiles_kernel(
    const uint32_t* __restrict__ req_offsets, uint32_t n_req, uint32_t* __restrict__ tile_req,
    uint32_t* __restrict__ tile_item0, uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ n_tiles,
    uint32_t* __restrict__ req_tile0, uint32_t tile_items) {
    __shared__ uint32_t chunk_sum[1024];
    __shared__ uint32_t pre[kTilesLdsReqs];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_req + 1023) / 1024;
    const uint32_t r0 = tid * per, r1 = (r0 + per < n_req) ? r0 + per : n_req;
    uint32_t sum = 0;
    for (uint32_t r = r0; r < r1; ++r) sum += (req_offsets[r + 1] - req_offsets[r] + tile_items - 1) / tile_items;
    chunk_sum[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {            // inclusive scan of the chunk sums
        const uint32_t v = tid >= d ? chunk_sum[tid - d] : 0;
        __syncthreads();
        chunk_sum[tid] += v;
        __syncthreads();
    }
    uint32_t acc = chunk_sum[tid] - sum;
    for (uint32_t r = r0; r < r1; ++r) {
        req_tile0[r] = acc;
        if (r < kTilesLdsReqs) pre[r] = acc;
        acc += (req_offsets[r + 1] - req_offsets[r] + tile_items - 1) / tile_items;
    }
    const uint32_t total = chunk_sum[1023];
    if (tid == 0) *n_tiles = total;
    __syncthreads();                                      // pre[] and (through L2, same block) req_tile0[] complete
    const bool in_lds = n_req <= kTilesLdsReqs;
    for (uint32_t t = tid; t < total; t += 1024) {
        // the last request whose first tile is <= t (requests without items share their successor's prefix and
        // are skipped by taking the LAST such entry)
        uint32_t lo = 0, hi = n_req - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            const uint32_t v = in_lds ? pre[mid] : req_tile0[mid];
            if (v <= t) lo = mid;
            else hi = mid - 1;
        }
        const uint32_t first = in_lds ? pre[lo] : req_tile0[lo];
        const uint32_t e = req_offsets[lo + 1];
        const uint32_t i = req_offsets[lo] + (t - first) * tile_items;
        tile_req[t] = lo;
        tile_item0[t] = i;
        tile_cnt[t] = (e - i < tile_items) ? e - i : tile_items;
    }
}

// The same table from many workgroups (n_req <= kTilesLdsReqs): every workgroup of 256 threads recomputes the request
// prefix for itself in LDS — a few loads per thread — and then writes ITS 256 tiles; workgroup 0 also publishes the
// prefix and the tile count.  One workgroup walking 20 000 tiles took 70-90 us of a 0.6 ms rank stage (r3 profile);
// this takes one short launch.  The grid covers the host's upper bound n_items / tile_items + n_req.
__device__ __forceinline__ void build_tiles_wide_body(
    uint32_t block, const uint32_t* __restrict__ req_offsets, uint32_t n_req, uint32_t* __restrict__ tile_req,
    uint32_t* __restrict__ tile_item0, uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ n_tiles,
    uint32_t* __restrict__ req_tile0, uint32_t tile_items) {
    __shared__ uint32_t chunk_sum[256];
    __shared__ uint32_t pre[kTilesLdsReqs];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_req + 255) / 256;
    const uint32_t r0 = tid * per < n_req ? tid * per : n_req, r1 = (r0 + per < n_req) ? r0 + per : n_req;
    uint32_t sum = 0;
    for (uint32_t r = r0; r < r1; ++r) sum += (req_offsets[r + 1] - req_offsets[r] + tile_items - 1) / tile_items;
    chunk_sum[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        const uint32_t v = tid >= d ? chunk_sum[tid - d] : 0;
        __syncthreads();
        chunk_sum[tid] += v;
        __syncthreads();
    }
    uint32_t acc = chunk_sum[tid] - sum;
    const bool publish = block == 0;
    for (uint32_t r = r0; r < r1; ++r) {
        pre[r] = acc;
        if (publish) req_tile0[r] = acc;
        acc += (req_offsets[r + 1] - req_offsets[r] + tile_items - 1) / tile_items;
    }
    const uint32_t total = chunk_sum[255];
    if (publish && tid == 0) *n_tiles = total;
    __syncthreads();
    const uint32_t t = block * 256 + tid;
    if (t >= total) return;
    uint32_t lo = 0, hi = n_req - 1;                       // the LAST request whose first tile is <= t (see above)
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (pre[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    const uint32_t e = req_offsets[lo + 1];
    const uint32_t i = req_offsets[lo] + (t - pre[lo]) * tile_items;
    tile_req[t] = lo;
    tile_item0[t] = i;
    tile_cnt[t] = (e - i < tile_items) ? e - i : tile_items;
}
__global__ __launch_bounds__(256) void build_tiles_wide_kernel(
    const uint32_t* __restrict__ req_offsets, uint32_t n_req, uint32_t* __restrict__ tile_req,
    uint32_t* __restrict__ tile_item0, uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ n_tiles,
    uint32_t* __restrict__ req_tile0, uint32_t tile_items) {
    build_tiles_wide_body(blockIdx.x, req_offsets, n_req, tile_req, tile_item0, tile_cnt, n_tiles, req_tile0, tile_items);
}
// the tile table as a third plane of the two-tower request-side launch (fm2t_user_fast_kernel, blockIdx.y == 2): the
// two are independent, and a launch boundary costs as much as the table does
struct TileTableArgs {
    const uint32_t* req_offsets;
    uint32_t n_req;
    uint32_t *tile_req, *tile_item0, *tile_cnt, *n_tiles, *req_tile0;
    uint32_t tile_items;
    uint32_t blocks;                                       // 0: no table in this launch
};

static int build_tiles_launch(pg_ctx* ctx, const uint32_t* d_off, uint32_t n_req, uint32_t max_tiles, uint32_t tile_items,
                              uint32_t* tile_req, uint32_t* tile_item0, uint32_t* tile_cnt, uint32_t* n_tiles,
                              uint32_t* req_tile0) {
    if (n_req <= kTilesLdsReqs)
        build_tiles_wide_kernel<<<(max_tiles + 255) / 256, 256, 0, ctx->stream>>>(d_off, n_req, tile_req, tile_item0, tile_cnt,
                                                                                  n_tiles, req_tile0, tile_items);
    else
        build_tiles_kernel<<<1, 1024, 0, ctx->stream>>>(d_off, n_req, tile_req, tile_item0, tile_cnt, n_tiles, req_tile0,
                                                        tile_items);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// DNN3 request-constant half of layer 1: c1[r][j] = chain(b1[j]; P(u[r][k]) * W1u[k][j], k asc)
// (W1u is stored already rounded to the model's operand precision).  The chain is sequential in k; its loads
// are not — 64 (then eight) rows of W1u are requested before the fmafs that use them.
__global__ void dnn3_user_partial_kernel(const float* __restrict__ user, uint32_t du,
                                         const float* __restrict__ w1u, const float* __restrict__ b1,
                                         uint32_t h1, int prec, float* __restrict__ c1) {
    const uint32_t r = blockIdx.x;
    const uint32_t j = blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= h1) return;
    const float* const u = user + (size_t)r * du;
    float acc = b1[j];
    uint32_t k = 0;
    // (64 rows per step: a lone request's launch is one round trip per step long — 16 steps of 8 rows took 14.8 us)
    for (; k + 64 <= du; k += 64) {
        float wv[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) wv[i] = w1u[(size_t)(k + i) * h1 + j];
#pragma unroll
        for (int i = 0; i < 64; ++i) acc = __fmaf_rn(round_prec(u[k + i], prec), wv[i], acc);
    }
    for (; k + 8 <= du; k += 8) {
        float wv[8], uv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            wv[i] = w1u[(size_t)(k + i) * h1 + j];
            uv[i] = round_prec(u[k + i], prec);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __fmaf_rn(uv[i], wv[i], acc);
    }
    for (; k < du; ++k) acc = __fmaf_rn(round_prec(u[k], prec), w1u[(size_t)k * h1 + j], acc);
    c1[(size_t)r * h1 + j] = acc;
}

// the FM prefix of a request (user fields first in the specification's sums), one workgroup beside the tower's
__device__ __forceinline__ void fm2t_user_prefix(uint32_t r, uint32_t tid, const float* const* __restrict__ field_emb,
                                                 const float* const* __restrict__ field_lin,
                                                 const int32_t* __restrict__ user_field_ids, uint32_t vocab, float fm_b,
                                                 float* __restrict__ fm_user, uint32_t nuf, uint32_t fk) {
    if (!user_field_ids) return;
    if (tid < fk) {
        float s = 0.0f, q = 0.0f;
        for (uint32_t f = 0; f < nuf; ++f) {
            int32_t id = user_field_ids[(size_t)r * nuf + f];
            id = id < 0 ? 0 : (id >= (int32_t)vocab ? (int32_t)vocab - 1 : id);
            const float v = field_emb[f][(size_t)id * fk + tid];
            s = s + v;
            q = __fmaf_rn(v, v, q);
        }
        fm_user[(size_t)r * kFmUserStride + 1 + tid] = s;
        fm_user[(size_t)r * kFmUserStride + 1 + kFmMaxK + tid] = q;
    }
    if (tid == 64) {
        float lin = fm_b;
        for (uint32_t f = 0; f < nuf; ++f) {
            int32_t id = user_field_ids[(size_t)r * nuf + f];
            id = id < 0 ? 0 : (id >= (int32_t)vocab ? (int32_t)vocab - 1 : id);
            lin = lin + field_lin[f][id];
        }
        fm_user[(size_t)r * kFmUserStride] = lin;
    }
}

// two-tower request side: user tower output uo[r][t_out] and the user prefix of the FM sums.
__global__ __launch_bounds__(256) void fm2t_user_kernel(
    const float* __restrict__ user, uint32_t du, const float* __restrict__ uw1,
    const float* __restrict__ ub1, const float* __restrict__ uw2, const float* __restrict__ ub2,
    uint32_t th, uint32_t to, int prec, const float* const* __restrict__ field_emb,
    const float* const* __restrict__ field_lin, const int32_t* __restrict__ user_field_ids,
    uint32_t vocab, float fm_b, float* __restrict__ uo, float* __restrict__ fm_user, uint32_t nuf, uint32_t fk) {
    __shared__ float u1[1024];
    __shared__ float us[4096];
    const uint32_t r = blockIdx.x, tid = threadIdx.x;
    if (blockIdx.y == 1) {                       // the FM prefix of the request, beside the tower (its own workgroup)
        fm2t_user_prefix(r, tid, field_emb, field_lin, user_field_ids, vocab, fm_b, fm_user, nuf, fk);
        return;
    }
    for (uint32_t k = tid; k < du; k += blockDim.x) us[k] = round_prec(user[(size_t)r * du + k], prec);
    __syncthreads();
    // the chains are sequential in k; their loads are not — eight weight rows are requested ahead of the eight fmafs
    for (uint32_t j = tid; j < th; j += blockDim.x) {
        float acc = ub1[j];
        uint32_t k = 0;
        for (; k + 8 <= du; k += 8) {
            float wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = uw1[(size_t)(k + i) * th + j];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = __fmaf_rn(us[k + i], wv[i], acc);
        }
        for (; k < du; ++k) acc = __fmaf_rn(us[k], uw1[(size_t)k * th + j], acc);
        u1[j] = round_prec(acc > 0.0f ? acc : 0.0f, prec);
    }
    __syncthreads();
    for (uint32_t o = tid; o < to; o += blockDim.x) {
        float acc = ub2[o];
        uint32_t j = 0;
        for (; j + 8 <= th; j += 8) {
            float wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = uw2[(size_t)(j + i) * to + o];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = __fmaf_rn(u1[j + i], wv[i], acc);
        }
        for (; j < th; ++j) acc = __fmaf_rn(u1[j], uw2[(size_t)j * to + o], acc);
        uo[(size_t)r * to + o] = acc;
    }
}

// The same tower when its shape allows every weight a thread needs to sit in its registers (user width DU, hidden width
// <= 256, 256 / t_out threads per output each owning SUB = t_hidden * t_out / 256 consecutive terms of the output's chain):
// ALL weight loads of both layers are issued before the first fmaf — one memory round trip for the request instead of
// du / 8 + t_hidden / 8 dependent ones (the generic kernel's layer 2 keeps only t_out threads busy: 32 round trips at
// 256 -> 64).  The chains themselves are unchanged (k ascending; an output's chain passes from part to part through LDS),
// so the results are the generic kernel's bit for bit.
template <int DU, int SUB>
__global__ __launch_bounds__(256) void fm2t_user_fast_kernel(
    const float* __restrict__ user, const float* __restrict__ uw1, const float* __restrict__ ub1,
    const float* __restrict__ uw2, const float* __restrict__ ub2, uint32_t th, uint32_t to, int prec,
    const float* const* __restrict__ field_emb, const float* const* __restrict__ field_lin,
    const int32_t* __restrict__ user_field_ids, uint32_t vocab, float fm_b, float* __restrict__ uo,
    float* __restrict__ fm_user, uint32_t nuf, uint32_t fk, uint32_t n_req, TileTableArgs tt) {
    __shared__ float us[DU];
    __shared__ float u1[256];
    __shared__ float carry[256];
    const uint32_t r = blockIdx.x, tid = threadIdx.x;
    if (blockIdx.y == 2) {
        if (r < tt.blocks)
            build_tiles_wide_body(r, tt.req_offsets, tt.n_req, tt.tile_req, tt.tile_item0, tt.tile_cnt, tt.n_tiles, tt.req_tile0, tt.tile_items);
        return;
    }
    if (r >= n_req) return;
    if (blockIdx.y == 1) {
        fm2t_user_prefix(r, tid, field_emb, field_lin, user_field_ids, vocab, fm_b, fm_user, nuf, fk);
        return;
    }
    const uint32_t parts = 256u / to, o = tid % to, part = tid / to;
    const bool l1 = tid < th;
    float w1v[DU], w2v[SUB];
    const float* w1c = uw1 + (l1 ? tid : 0u);
#pragma unroll
    for (int k = 0; k < DU; ++k) w1v[k] = w1c[(size_t)k * th];
    const float* w2c = uw2 + (size_t)(part * SUB) * to + o;
#pragma unroll
    for (int i = 0; i < SUB; ++i) w2v[i] = w2c[(size_t)i * to];
    float acc = ub1[l1 ? tid : 0u];
    float acc2 = ub2[o];
    if (tid < DU) us[tid] = round_prec(user[(size_t)r * DU + tid], prec);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DU; ++k) acc = __fmaf_rn(us[k], w1v[k], acc);
    if (l1) u1[tid] = round_prec(acc > 0.0f ? acc : 0.0f, prec);
    __syncthreads();
    for (uint32_t p = 0; p < parts; ++p) {
        if (part == p) {
            if (p) acc2 = carry[o];
#pragma unroll
            for (int i = 0; i < SUB; ++i) acc2 = __fmaf_rn(u1[p * SUB + i], w2v[i], acc2);
            if (p + 1 == parts) uo[(size_t)r * to + o] = acc2;
            else carry[o] = acc2;
        }
        __syncthreads();
    }
}

// launches the request side of FM + two-tower: the register-resident form where the shape allows it.  `tt` (optional): a
// tile table to build in the same launch; *tiles_built says whether that happened (else the caller launches the table)
static void launch_fm2t_user(pg_ctx* ctx, const pg_model* m, const float* d_user, const int32_t* d_ufids, uint32_t n_req,
                             bool with_prefix, float* uo, float* fm_user, const TileTableArgs* tt = nullptr, bool* tiles_built = nullptr) {
    if (tiles_built) *tiles_built = false;
    const float* const* fe = with_prefix ? m->d_field_emb : nullptr;
    const float* const* fl = with_prefix ? m->d_field_lin : nullptr;
    const uint32_t sub = (m->to && 256u % m->to == 0) ? m->th * m->to / 256u : 0;
    const bool fits = m->d_user == 128 && m->th <= 256 && m->to <= 256 && sub * (256u / (m->to ? m->to : 1)) == m->th &&
                      !ctx->knobs.rank_no_ws;
    if (fits && (sub == 64 || sub == 32)) {
        TileTableArgs t{};
        if (tt && with_prefix && tt->blocks && tt->n_req <= kTilesLdsReqs) {
            t = *tt;
            if (tiles_built) *tiles_built = true;
        }
        const dim3 grid(std::max(n_req, t.blocks), t.blocks ? 3 : (with_prefix ? 2 : 1));
        if (sub == 64)
            fm2t_user_fast_kernel<128, 64><<<grid, 256, 0, ctx->stream>>>(d_user, m->w1u, m->b1, m->uw2, m->ub2, m->th, m->to, m->prec, fe, fl,
                                                                         d_ufids, m->vocab, m->fm_b, uo, fm_user, m->nuf, m->k, n_req, t);
        else
            fm2t_user_fast_kernel<128, 32><<<grid, 256, 0, ctx->stream>>>(d_user, m->w1u, m->b1, m->uw2, m->ub2, m->th, m->to, m->prec, fe, fl,
                                                                         d_ufids, m->vocab, m->fm_b, uo, fm_user, m->nuf, m->k, n_req, t);
    } else {
        fm2t_user_kernel<<<dim3(n_req, with_prefix ? 2 : 1), 256, 0, ctx->stream>>>(d_user, m->d_user, m->w1u, m->b1, m->uw2, m->ub2, m->th, m->to,
                                                                                    m->prec, fe, fl, d_ufids, m->vocab, m->fm_b, uo, fm_user, m->nuf, m->k);
    }
}

}  // namespace pg

// ---------------------------------------------------------------------------------------------
// host side: model blobs, weight pre-packing, launches
// ---------------------------------------------------------------------------------------------
namespace pg {

// Pack W[K][N] (row-major, k major) into MFMA B-fragment order, 1 KiB per (n-block, k-group).
//   bf16: fragment (nbg, ks): lane (j,hk) holds W[ks*16 + 8*hk + e][nbg*32 + j], e = 0..7
//   f32 : fragment (nbg, g8): lane (j,h)  holds W[g8*8 + 2*st + h][nbg*32 + j], st = 0..3
//   split bf16 (prec 2): the bf16 layout twice — every hi fragment, then every lo fragment (lo = RNE(w - hi))
static std::vector<uint8_t> pack_weights(const float* w, uint32_t K, uint32_t N, int prec) {
    const uint32_t kg = prec ? K / 16 : K / 8;
    const size_t plane = (size_t)(N / 32) * kg * 1024;
    std::vector<uint8_t> out(plane * (prec == 2 ? 2 : 1));
    for (uint32_t nbg = 0; nbg < N / 32; ++nbg)
        for (uint32_t g = 0; g < kg; ++g) {
            uint8_t* frag = out.data() + ((size_t)nbg * kg + g) * 1024;
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t j = lane & 31, hh = lane >> 5;
                if (prec) {
                    uint16_t* d = reinterpret_cast<uint16_t*>(frag + lane * 16);
                    uint16_t* dl = reinterpret_cast<uint16_t*>(frag + plane + lane * 16);
                    for (uint32_t e = 0; e < 8; ++e) {
                        const float v = w[(size_t)(g * 16 + 8 * hh + e) * N + nbg * 32 + j];
                        d[e] = f32_to_bf16_rne(v);
                        if (prec == 2) dl[e] = f32_to_bf16_rne(v - bf16_to_f32(d[e]));
                    }
                } else {
                    float* d = reinterpret_cast<float*>(frag + lane * 16);
                    for (uint32_t st = 0; st < 4; ++st)
                        d[st] = w[(size_t)(g * 8 + 2 * st + hh) * N + nbg * 32 + j];
                }
            }
        }
    return out;
}

static int upload(pg_ctx* ctx, pg_model* m, const void* src, size_t bytes, void** dst) {
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, bytes ? bytes : 16);
    if (e != hipSuccess) {
        set_error("pg_model_load: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return PG_ERR_NOMEM;
    }
    m->allocs.push_back(d);
    if (bytes) PG_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    *dst = d;
    return PG_OK;
}

static std::vector<float> rounded(const float* w, size_t n, int prec) {
    std::vector<float> o(n);
    for (size_t i = 0; i < n; ++i) o[i] = round_prec(w[i], prec);
    return o;
}


struct RankScratch {
    uint32_t *tile_req, *tile_item0, *tile_cnt, *n_tiles, *req_tile0;
    float* c1;       // [n_req][h1]  (DNN3) or uo [n_req][to] (two-tower)
    float* fm_user;  // [n_req][kFmUserStride]
    float* sink;     // [1024] write-only
};

static int rank_scratch(pg_ctx* ctx, uint32_t n_req, uint32_t max_tiles, uint32_t per_req_floats,
                        RankScratch* rs) {
    void* p;
    int rc;
    const size_t ints = (size_t)3 * max_tiles + 64 + n_req;
    const size_t bytes = ints * 4 + ((size_t)n_req * per_req_floats + (size_t)n_req * kFmUserStride) * 4 + 256 + 8192;
    if ((rc = scratch_reserve(ctx, 6, bytes, &p))) return rc;
    uint32_t* u = (uint32_t*)p;
    rs->tile_req = u;
    rs->tile_item0 = u + max_tiles;
    rs->tile_cnt = u + 2 * (size_t)max_tiles;
    rs->n_tiles = u + 3 * (size_t)max_tiles;
    rs->req_tile0 = rs->n_tiles + 64;
    rs->c1 = (float*)(rs->req_tile0 + n_req);
    rs->fm_user = rs->c1 + (size_t)n_req * per_req_floats;
    rs->sink = rs->fm_user + (size_t)n_req * kFmUserStride + 16;     // 1 024 floats nobody reads (rank_ir.hip's always-issued stores)
    return PG_OK;
}

// ---- shapes the fused kernel is instantiated for ----------------------------------------------------------------
// Tiling per precision.  bf16: 2 x 2 waves, 64-column layer-1 chunks, two-pass head, two workgroups
// per CU with pre-loaded B fragments while the accumulators fit 256 registers (h2 <= 256), one workgroup per CU beyond;
// fp32: 1 x 4 waves, 128-column chunks, head in one pass (two when the fp32 H2 tile would not fit the LDS).
template <int PREC, int H1, int H2>
static int launch_dnn3_mlp(pg_ctx* ctx, const MlpArgs& a, uint32_t grid) {
    int rc;
    if constexpr (PREC == 1) {
        constexpr int OCC = H2 <= 256 ? 2 : 1;
        constexpr size_t lds = mlp_lds_bytes(1, H2, 64, 2);
        if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<1, H1, H2, true, 2, 2, 1, 64, 2, OCC>, lds))) return rc;
        mlp_kernel<1, H1, H2, true, 2, 2, 1, 64, 2, OCC><<<grid, 256, lds, ctx->stream>>>(a);
    } else if constexpr (PREC == 2) {
        // split bf16, the general form (the benchmark's shape has a kernel of its own, rank_x3.hip): the bf16 tiling with
        // a hi and a lo operand tile, one workgroup per CU's worth of registers
        constexpr size_t lds = mlp_lds_bytes(2, H2, 64, 2);
        if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<2, H1, H2, true, 2, 2, 1, 64, 2, 1>, lds))) return rc;
        mlp_kernel<2, H1, H2, true, 2, 2, 1, 64, 2, 1><<<grid, 256, lds, ctx->stream>>>(a);
    } else {
        constexpr int EP = H2 > 256 ? 2 : 1;
        constexpr size_t lds = mlp_lds_bytes(0, H2, 128, EP);
        if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<0, H1, H2, true, 1, 4, 1, 128, EP, 1>, lds))) return rc;
        mlp_kernel<0, H1, H2, true, 1, 4, 1, 128, EP, 1><<<grid, 256, lds, ctx->stream>>>(a);
    }
    return PG_OK;
}
// two-tower bf16 tiles.  64-item tiles (four workgroups per CU, no pre-loaded B fragments) measured SLOWER than 128-item
// tiles with two workgroups per CU and pre-loaded fragments (0.66 vs 0.52 ms per 1.28 M items): every tile re-reads the
// towers' 96 KB of weight fragments from L2, so halving the tile doubles that traffic — the kernel wants its weights
// stationary, not more occupancy
constexpr int kFmBM = 128;
template <int PREC, int TH, int TO, int FK>
static int launch_fm2t_mlp(pg_ctx* ctx, const MlpArgs& a, uint32_t grid) {
    int rc;
    if (a.irows) {               // candidates come as rows of materialised item records (MODEL 3)
        if constexpr (PREC == 1) {
            constexpr size_t lds = mlp_lds_bytes(1, TO, 128, 1, kFmBM);
            if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<1, TH, TO, false, 2, 2, 3, 128, 1, 2, FK, kFmBM>, lds))) return rc;
#ifdef PG_MLP_PROFILE
            // developer aid (make MLP_EXTRA=-DPG_MLP_PROFILE): per-phase cycle counts of 16 mid-grid workgroups, printed once
            static uint64_t* dbg = nullptr;
            if (!dbg) (void)hipMalloc(&dbg, 16 * 4 * 8 * 8);
            MlpArgs b = a;
            b.field_emb = reinterpret_cast<const float* const*>(dbg);
            mlp_kernel<1, TH, TO, false, 2, 2, 3, 128, 1, 2, FK, kFmBM><<<grid, 256, lds, ctx->stream>>>(b);
            uint64_t hcyc[16 * 4 * 8];
            (void)hipMemcpy(hcyc, dbg, sizeof hcyc, hipMemcpyDeviceToHost);
            static int calls = 0;
            if (++calls == 5) {
                double av[8] = {0};
                for (int w = 0; w < 64; ++w)
                    for (int i = 0; i < 8; ++i) av[i] += (double)hcyc[w * 8 + i] / 64.0;
                fprintf(stderr, "mlp model 3, mean cycles per wave and tile: gather+fm %.0f | barrier %.0f | L1 mfma %.0f | relu+store %.0f | barrier %.0f | L2 mfma %.0f | barrier %.0f | head %.0f\n",
                        av[0], av[1], av[2], av[3], av[4], av[5], av[6], av[7]);
            }
            return PG_OK;
#endif
            mlp_kernel<1, TH, TO, false, 2, 2, 3, 128, 1, 2, FK, kFmBM><<<grid, 256, lds, ctx->stream>>>(a);
        } else if constexpr (PREC == 2) {
            constexpr size_t lds = mlp_lds_bytes(2, TO, 128, 1, kFmBM);
            if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<2, TH, TO, false, 2, 2, 3, 128, 1, 1, FK, kFmBM>, lds))) return rc;
            mlp_kernel<2, TH, TO, false, 2, 2, 3, 128, 1, 1, FK, kFmBM><<<grid, 256, lds, ctx->stream>>>(a);
        } else {
            constexpr size_t lds = mlp_lds_bytes(0, TO, 128, 1);
            if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<0, TH, TO, false, 2, 2, 3, 128, 1, 1, FK>, lds))) return rc;
            mlp_kernel<0, TH, TO, false, 2, 2, 3, 128, 1, 1, FK><<<grid, 256, lds, ctx->stream>>>(a);
        }
        return PG_OK;
    }
    if constexpr (PREC == 1) {
        // (three workgroups per CU — 64-column layer-1 chunks, 48 KB of LDS, <= 168 registers — measured 0.53 vs 0.45 ms:
        //  twice the chunks and barriers per tile cost more than the extra waves hide)
        constexpr size_t lds = mlp_lds_bytes(1, TO, 128, 1, kFmBM);
        if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<1, TH, TO, false, 2, 2, 2, 128, 1, 2, FK, kFmBM>, lds))) return rc;
        mlp_kernel<1, TH, TO, false, 2, 2, 2, 128, 1, 2, FK, kFmBM><<<grid, 256, lds, ctx->stream>>>(a);
    } else if constexpr (PREC == 2) {
        constexpr size_t lds = mlp_lds_bytes(2, TO, 128, 1, kFmBM);
        if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<2, TH, TO, false, 2, 2, 2, 128, 1, 1, FK, kFmBM>, lds))) return rc;
        mlp_kernel<2, TH, TO, false, 2, 2, 2, 128, 1, 1, FK, kFmBM><<<grid, 256, lds, ctx->stream>>>(a);
    } else {
        constexpr size_t lds = mlp_lds_bytes(0, TO, 128, 1);
        if ((rc = ensure_dyn_lds(ctx, (const void*)mlp_kernel<0, TH, TO, false, 2, 2, 2, 128, 1, 1, FK>, lds))) return rc;
        mlp_kernel<0, TH, TO, false, 2, 2, 2, 128, 1, 1, FK><<<grid, 256, lds, ctx->stream>>>(a);
    }
    return PG_OK;
}
// (h1, h2) of DNN3 and (t_h1, t_out, k) of the two-tower model; d_item is 64 or 128 (64: the gathered row is padded
// with zero columns whose W1 rows are zero), n_item_fields x k = 128, n_user_fields <= 16
#define PG_DNN3_SHAPES(X) X(128, 128) X(256, 128) X(256, 256) X(512, 256) X(1024, 512)
#define PG_FM2T_SHAPES(X) X(256, 64, 16) X(256, 64, 32) X(128, 64, 8) X(512, 128, 16)
static bool dnn3_shape_ok(uint32_t h1, uint32_t h2) {
#define X(A, B) if (h1 == A && h2 == B) return true;
    PG_DNN3_SHAPES(X)
#undef X
    return false;
}
static bool fm2t_shape_ok(uint32_t th, uint32_t to, uint32_t k) {
#define X(A, B, C) if (th == A && to == B && k == C) return true;
    PG_FM2T_SHAPES(X)
#undef X
    return false;
}
static int dispatch_dnn3_mlp(pg_ctx* ctx, const pg_model* m, const MlpArgs& a, uint32_t grid) {
#define X(A, B)                                                                                          \
    if (m->h1 == A && m->h2 == B)                                                                        \
        return m->prec == 2 ? launch_dnn3_mlp<2, A, B>(ctx, a, grid)                                      \
                            : (m->prec ? launch_dnn3_mlp<1, A, B>(ctx, a, grid) : launch_dnn3_mlp<0, A, B>(ctx, a, grid));
    PG_DNN3_SHAPES(X)
#undef X
    set_error("rank: DNN3 shape ->%u->%u has no kernel", m->h1, m->h2);
    return PG_ERR_UNSUPPORTED;
}
static int dispatch_fm2t_mlp(pg_ctx* ctx, const pg_model* m, const MlpArgs& a, uint32_t grid) {
#define X(A, B, C)                                                                                       \
    if (m->th == A && m->to == B && m->k == C)                                                           \
        return m->prec == 2 ? launch_fm2t_mlp<2, A, B, C>(ctx, a, grid)                                   \
                            : (m->prec ? launch_fm2t_mlp<1, A, B, C>(ctx, a, grid) : launch_fm2t_mlp<0, A, B, C>(ctx, a, grid));
    PG_FM2T_SHAPES(X)
#undef X
    set_error("rank: two-tower shape ->%u->%u, k=%u has no kernel", m->th, m->to, m->k);
    return PG_ERR_UNSUPPORTED;
}

int rank_dnn3_dev_locked(pg_ctx* ctx, const pg_model* m, const pg_table* t,
                                const float* d_user, const uint32_t* d_cand, const uint32_t* d_off,
                                uint32_t n_req, uint32_t n_items, float* d_out, size_t out_stride) {
    if (n_items == 0 || n_req == 0) return PG_OK;
    const uint32_t max_tiles = n_items / 32 + n_req;            // sized for the smallest (32-item) tiles
    RankScratch rs;
    int rc;
    if ((rc = rank_scratch(ctx, n_req, max_tiles, m->h1, &rs))) return rc;
    const bool no_ws = ctx->knobs.rank_no_ws;        // A/B switch: the streaming kernel
    // the weights-stationary kernel is built for the benchmark's shape: [d_user + 128] -> 512 -> 256 -> 1 in bf16
    const bool ws = m->prec == 1 && !no_ws && m->h1 == 512 && m->h2 == 256 && t->dim == 128;
    // the small shapes, gather-bound: the whole model in registers (rank_rs.hip)
    const bool rs_k = m->prec == 1 && !no_ws && t->dim == 128 && dnn3_rs_shape(m->h1, m->h2);
    const bool ls_k = m->prec == 1 && !no_ws && t->dim == 128 && dnn3_ls_shape(m->h1, m->h2);
    // split bf16: the two-role kernel (rank_x3.hip), 128-item tiles; 1024-512 and 64-wide tables take the general form
    const bool x3_k = m->prec == 2 && !no_ws && t->dim == 128 && dnn3_x3_shape(m->h1, m->h2);
    const uint32_t grid128 = n_items / kBM + n_req;
    if (!ctx->timers_off) PG_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    if ((rc = build_tiles_launch(ctx, d_off, n_req, max_tiles, ws || rs_k ? (uint32_t)kWsItems : (uint32_t)kBM, rs.tile_req, rs.tile_item0,
                                 rs.tile_cnt, rs.n_tiles, rs.req_tile0)))
        return rc;
    dnn3_user_partial_kernel<<<dim3(n_req, (m->h1 + 255) / 256), 256, 0, ctx->stream>>>(
        d_user, m->d_user, m->w1u, m->b1, m->h1, m->prec, rs.c1);
    MlpArgs a{};
    a.tile_req = rs.tile_req;
    a.tile_item0 = rs.tile_item0;
    a.tile_cnt = rs.tile_cnt;
    a.n_tiles = rs.n_tiles;
    a.tab = t->d;
    a.tab_rows = (uint32_t)t->rows;
    a.tab_dim = t->dim;
    a.cand_rows = d_cand;
    a.c1 = rs.c1;
    a.c1_stride = m->h1;
    a.w3 = m->w3;
    a.w3_stride = 0;
    a.b3 = m->b3;
    a.b2 = m->b2;
    a.w1p = m->w1p;
    a.w2p = m->w2p;
    a.w1p_lo = m->w1p_lo;
    a.w2p_lo = m->w2p_lo;
    a.out = d_out;
    a.n_out = m->n_out;
    a.out_stride = out_stride ? out_stride : (size_t)n_items;
    a.b3v = m->b3v;
    a.head_part = nullptr;
    if (ws && m->n_out > 1) {
        // the weights-stationary kernel's partials of heads 1..: a block per workgroup (its LDS is full)
        void* hp;
        if ((rc = scratch_reserve(ctx, 14, (size_t)ctx->num_cus * (kMaxHeads - 1) * 4 * kWsItems * 4, &hp))) return rc;
        a.head_part = (float*)hp;
    }
    if (ws) {
        // bf16: weights-stationary persistent kernel over 64-item tiles
        if ((rc = launch_dnn3_ws(ctx, a))) return rc;
    } else if (rs_k) {
        if ((rc = launch_dnn3_rs(ctx, m->h1, m->h2, a))) return rc;
    } else if (ls_k) {
        if ((rc = launch_dnn3_ls(ctx, m->h1, m->h2, a))) return rc;
    } else if (x3_k) {
        if ((rc = launch_dnn3_x3(ctx, m->h1, m->h2, a))) return rc;
    } else if ((rc = dispatch_dnn3_mlp(ctx, m, a, grid128))) {
        return rc;
    }
    PG_HIP(hipGetLastError());
    if (!ctx->timers_off) {
        PG_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
        ctx->rank_timing_pending = true;
    }
    ctx->stats.rank_calls++;
    ctx->stats.rank_items += n_items;
    return PG_OK;
}

static int rank_fm2t_dev_locked(pg_ctx* ctx, const pg_model* m, const float* d_user,
                                const int32_t* d_ufids, const int32_t* d_ifids, const uint32_t* d_off,
                                uint32_t n_req, uint32_t n_items, float* d_out, const pg_item_rows* ir = nullptr,
                                const uint32_t* d_cand = nullptr) {
    if (n_items == 0 || n_req == 0) return PG_OK;
    // the benchmark's shape over item records: the stationary-weights kernel (rank_ir.hip), 64-item tiles
    const bool irs = ir && !ctx->knobs.rank_no_ws && fm2t_irs_shape(m->th, m->to, m->k, m->nif, m->prec);
    const bool isw = irs && !ctx->knobs.fm2t_irs;
    const uint32_t bm = isw ? (uint32_t)kIswItems : irs ? (uint32_t)kIrsItems : (m->prec ? (uint32_t)kFmBM : (uint32_t)kBM);
    const uint32_t max_tiles = n_items / bm + n_req;
    RankScratch rs;
    int rc;
    if ((rc = rank_scratch(ctx, n_req, max_tiles, m->to, &rs))) return rc;
    if (!ctx->timers_off) PG_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
    // the request side and the tile table: one launch where the register-resident user kernel serves the shape
    const TileTableArgs tt{d_off, n_req, rs.tile_req, rs.tile_item0, rs.tile_cnt, rs.n_tiles, rs.req_tile0, bm, (max_tiles + 255) / 256};
    bool tiles_built = false;
    launch_fm2t_user(ctx, m, d_user, d_ufids, n_req, true, rs.c1, rs.fm_user, &tt, &tiles_built);
    if (!tiles_built && (rc = build_tiles_launch(ctx, d_off, n_req, max_tiles, bm, rs.tile_req, rs.tile_item0, rs.tile_cnt, rs.n_tiles,
                                                 rs.req_tile0)))
        return rc;
    MlpArgs a{};
    a.tile_req = rs.tile_req;
    a.tile_item0 = rs.tile_item0;
    a.tile_cnt = rs.tile_cnt;
    a.n_tiles = rs.n_tiles;
    a.field_emb = m->d_field_emb;
    a.field_lin = m->d_field_lin;
    a.item_field_ids = d_ifids;
    if (ir) {
        a.irows = ir->d;
        a.irow_count = (uint32_t)ir->rows;
        a.cand_rows = d_cand;
    }
    a.vocab = m->vocab;
    a.n_user_fields = m->nuf;
    a.fm_user = rs.fm_user;
    a.c1 = m->c1_shared;
    a.c1_stride = 0;
    a.w3 = rs.c1;            // user-tower output per request
    a.w3_stride = m->to;
    a.b2 = m->b2;
    a.w1p = m->w1p;
    a.w2p = m->w2p;
    a.w1p_lo = m->w1p_lo;
    a.w2p_lo = m->w2p_lo;
    a.out = d_out;
    a.sink = rs.sink;
    if (isw) {
        if ((rc = launch_fm2t_isw(ctx, a))) return rc;
    } else if (irs) {
        if ((rc = launch_fm2t_irs(ctx, a))) return rc;
    } else if ((rc = dispatch_fm2t_mlp(ctx, m, a, max_tiles))) {
        return rc;
    }
    PG_HIP(hipGetLastError());
    if (!ctx->timers_off) {
        PG_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
        ctx->rank_timing_pending = true;
    }
    ctx->stats.rank_calls++;
    ctx->stats.rank_items += n_items;
    return PG_OK;
}

// ---- materialised item records --------------------------------------------------------------------------------
// An item's field ids are static, so its side of the model can be laid out once: record r = the embeddings of the ids in
// row r of the item-field columns, concatenated in field order (kDIN floats), followed by their linear weights — ONE
// contiguous gather per candidate at rank time instead of the id row plus n_item_fields scattered embedding rows.  Record
// `rows` (one past the last item) holds the columns' defaults: what a candidate outside the feature store reads.
__global__ void item_rows_build_kernel(ItemRowCols cols, const float* const* __restrict__ field_emb,
                                       const float* const* __restrict__ field_lin, uint32_t nuf, uint32_t nif, uint32_t fk,
                                       uint32_t vocab, uint64_t fs_rows, uint64_t row0, uint64_t nrows, float* __restrict__ out) {
    // one thread per 16-B quad of a record: quads [0, 32) embeddings, [32, 40) linear weights + padding
    constexpr uint32_t QPR = kItemRowFloats / 4;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = gid / QPR;
    const uint32_t qd = (uint32_t)(gid % QPR);
    if (i >= nrows) return;
    const uint64_t row = row0 + i;
    auto id_of = [&](uint32_t f) -> int32_t {
        int64_t v;
        if (row >= fs_rows) v = (int64_t)cols.def[f];
        else v = cols.dtype[f] == PG_F_I32 ? (int64_t)((const int32_t*)cols.base[f])[row] : ((const int64_t*)cols.base[f])[row];
        v = v > 2147483647ll ? 2147483647ll : (v < -2147483648ll ? -2147483648ll : v);      // as features_gather_i32
        int32_t id = (int32_t)v;
        return id < 0 ? 0 : (id >= (int32_t)vocab ? (int32_t)vocab - 1 : id);                // as the per-field gather clamps
    };
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qd < kDIN / 4) {
        const uint32_t f = qd * 4 / fk, c0 = qd * 4 % fk;
        o = *reinterpret_cast<const float4*>(field_emb[nuf + f] + (size_t)id_of(f) * fk + c0);
    } else {
        float l[4] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t f = (qd - kDIN / 4) * 4 + j;
            if (f < nif) l[j] = field_lin[nuf + f][id_of(f)];
        }
        o = make_float4(l[0], l[1], l[2], l[3]);
    }
    *reinterpret_cast<float4*>(out + row * kItemRowFloats + 4 * qd) = o;
}

int item_rows_fill_locked(pg_ctx* ctx, pg_item_rows* ir, uint64_t row0, uint64_t nrows) {
    const pg_model* m = ir->m;
    ItemRowCols cols;
    memset(&cols, 0, sizeof cols);
    for (uint32_t f = 0; f < m->nif; ++f) {
        const int32_t ci = ir->cols[f];
        if (ci < 0 || (size_t)ci >= ir->fs->cols.size()) {
            set_error("pg_fm2t_item_rows: column index %d out of range (%zu columns)", ci, ir->fs->cols.size());
            return PG_ERR_INVALID;
        }
        const auto& c = ir->fs->cols[(size_t)ci];
        if (c.dtype != PG_F_I32 && c.dtype != PG_F_I64) {
            set_error("pg_fm2t_item_rows: column \"%s\" is not an integer column", c.name.c_str());
            return PG_ERR_INVALID;
        }
        cols.base[f] = c.d;
        cols.dtype[f] = c.dtype;
        cols.def[f] = c.def;
    }
    if (nrows == 0) return PG_OK;
    const uint64_t threads = nrows * (kItemRowFloats / 4);
    item_rows_build_kernel<<<(uint32_t)((threads + 255) / 256), 256, 0, ctx->stream>>>(cols, m->d_field_emb, m->d_field_lin, m->nuf, m->nif, m->k,
                                                                                      m->vocab, ir->fs->rows, row0, nrows, ir->d);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

int rank_fm2t_irows_dev_locked(pg_ctx* ctx, const pg_model* m, const pg_item_rows* ir, const float* d_user, const int32_t* d_ufids,
                               const uint32_t* d_cand, const uint32_t* d_off, uint32_t n_req, uint32_t n_items, float* d_out) {
    return rank_fm2t_dev_locked(ctx, m, d_user, d_ufids, nullptr, d_off, n_req, n_items, d_out, ir, d_cand);
}

// FM + two-tower straight from candidate rows: the per-item field ids are assembled on the device from the feature
// columns (no host boxing), then ranked.  Caller holds ctx->mu; nothing synchronises (<= 16 item fields).
int rank_fm2t_rows_dev_locked(pg_ctx* ctx, const pg_model* m, const pg_features* fs, const int32_t* item_field_cols,
                              const float* d_user, const int32_t* d_ufids, const uint32_t* d_cand, const uint32_t* d_off,
                              uint32_t n_req, uint32_t n_items, float* d_out) {
    if (n_items == 0 || n_req == 0) return PG_OK;
    void* ids;
    int rc;
    if ((rc = scratch_reserve(ctx, 0, (size_t)n_items * m->nif * 4, &ids))) return rc;
    if ((rc = features_gather_i32_locked(ctx, fs, item_field_cols, m->nif, d_cand, n_items, (int32_t*)ids, "rank_fm2t_rows"))) return rc;
    return rank_fm2t_dev_locked(ctx, m, d_user, d_ufids, (const int32_t*)ids, d_off, n_req, n_items, d_out);
}

// user-tower output uo[r][t_out] only: the "user embedding" an EasyRec / TorchRec vector model serves
int fm2t_user_embedding_locked(pg_ctx* ctx, const pg_model* m, const float* d_user, uint32_t n_req, float* d_out) {
    if (n_req == 0) return PG_OK;
    launch_fm2t_user(ctx, m, d_user, nullptr, n_req, false, d_out, nullptr);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

static int finish_rank_timing(pg_ctx* ctx) {
    PG_HIP(hipStreamSynchronize(ctx->stream));
    if (!ctx->rank_timing_pending) return PG_OK;       // (stage timers off: nothing was recorded, last_rank_ms stays)
    float ms = 0.f;
    PG_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
    ctx->stats.last_rank_ms = ms;
    ctx->rank_timing_pending = false;
    return PG_OK;
}

}  // namespace pg

extern "C" {

int pg_model_load(pg_ctx* ctx, pg_model_kind kind, pg_prec prec, const void* blob, size_t len,
                  pg_model** out) {
    PG_REQUIRE(ctx && blob && out, "pg_model_load: NULL argument");
    PG_REQUIRE(prec == PG_PREC_F32 || prec == PG_PREC_BF16 || prec == PG_PREC_BF16X3, "pg_model_load: bad precision %d", (int)prec);
    std::lock_guard<std::mutex> g(ctx->mu);
    PG_HIP(hipSetDevice(ctx->device));
    const uint8_t* p = (const uint8_t*)blob;
    pg_model* m = new pg_model();
    m->kind = kind;
    m->prec = (int)prec;
    int rc = PG_OK;
    auto fail = [&](int code) {
        for (void* a : m->allocs) hipFree(a);
        delete m;
        return code;
    };
    if (kind == PG_MODEL_DNN3 || kind == PG_MODEL_DNN3_MULTI) {
        const bool multi = kind == PG_MODEL_DNN3_MULTI;
        const size_t hb = multi ? 20 : 16;
        m->kind = PG_MODEL_DNN3;                 // every rank entry point takes it; n_out says how many planes it writes
        if (len < hb) { pg::set_error("pg_model_load: blob too short"); return fail(PG_ERR_INVALID); }
        uint32_t hdr[5] = {0, 0, 0, 0, 1};
        memcpy(hdr, p, hb);
        m->d_user = hdr[0]; m->d_item = hdr[1]; m->h1 = hdr[2]; m->h2 = hdr[3]; m->n_out = hdr[4];
        if (m->n_out == 0 || m->n_out > (uint32_t)pg::kMaxHeads) {
            pg::set_error("pg_model_load: a multi-output DNN3 has 1..%d outputs, the blob says %u", pg::kMaxHeads, m->n_out);
            return fail(PG_ERR_UNSUPPORTED);
        }
        if ((m->d_item != 128 && m->d_item != 64) || !pg::dnn3_shape_ok(m->h1, m->h2) || m->d_user == 0 || m->d_user > 4096) {
            pg::set_error("pg_model_load: DNN3 shape [%u+%u]->%u->%u->1 unsupported (d_item 64 or 128; hidden widths "
                          "128-128, 256-128, 256-256, 512-256, 1024-512)", m->d_user, m->d_item, m->h1, m->h2);
            return fail(PG_ERR_UNSUPPORTED);
        }
        const size_t din = (size_t)m->d_user + m->d_item;
        const size_t need = hb + (din * m->h1 + m->h1 + (size_t)m->h1 * m->h2 + m->h2 + ((size_t)m->h2 + 1) * m->n_out) * 4;
        if (len != need) { pg::set_error("pg_model_load: DNN3 blob is %zu bytes, expected %zu", len, need); return fail(PG_ERR_INVALID); }
        const float* w1 = (const float*)(p + hb);
        const float* b1 = w1 + din * m->h1;
        const float* w2 = b1 + m->h1;
        const float* b2 = w2 + (size_t)m->h1 * m->h2;
        const float* w3 = b2 + m->h2;                      // [h2][n_out] as exported ([in][out]); kept head-major on the device
        const float* b3 = w3 + (size_t)m->h2 * m->n_out;
        m->b3 = b3[0];
        std::vector<float> w3t((size_t)m->n_out * m->h2);
        for (uint32_t o = 0; o < m->n_out; ++o)
            for (uint32_t j = 0; j < m->h2; ++j) w3t[(size_t)o * m->h2 + j] = w3[(size_t)j * m->n_out + o];
        auto w1u = pg::rounded(w1, (size_t)m->d_user * m->h1, m->prec);
        // the kernel's layer 1 is 128 deep: a 64-wide item row is padded with zero columns, W1 with zero rows
        std::vector<float> w1i((size_t)pg::kDIN * m->h1, 0.0f);
        memcpy(w1i.data(), w1 + (size_t)m->d_user * m->h1, (size_t)m->d_item * m->h1 * 4);
        auto w1p = pg::pack_weights(w1i.data(), pg::kDIN, m->h1, m->prec);
        auto w2p = pg::pack_weights(w2, m->h1, m->h2, m->prec);
        if ((rc = pg::upload(ctx, m, w1u.data(), w1u.size() * 4, (void**)&m->w1u))) return fail(rc);
        if ((rc = pg::upload(ctx, m, b1, m->h1 * 4, (void**)&m->b1))) return fail(rc);
        if ((rc = pg::upload(ctx, m, w1p.data(), w1p.size(), &m->w1p))) return fail(rc);
        if ((rc = pg::upload(ctx, m, w2p.data(), w2p.size(), &m->w2p))) return fail(rc);
        if (m->prec == 2) {                          // split bf16: the lo fragments follow the hi fragments
            m->w1p_lo = (char*)m->w1p + w1p.size() / 2;
            m->w2p_lo = (char*)m->w2p + w2p.size() / 2;
        }
        if ((rc = pg::upload(ctx, m, b2, m->h2 * 4, (void**)&m->b2))) return fail(rc);
        if ((rc = pg::upload(ctx, m, w3t.data(), w3t.size() * 4, (void**)&m->w3))) return fail(rc);
        if ((rc = pg::upload(ctx, m, b3, m->n_out * 4, (void**)&m->b3v))) return fail(rc);
    } else if (kind == PG_MODEL_FM_TWOTOWER) {
        if (len < 32) { pg::set_error("pg_model_load: blob too short"); return fail(PG_ERR_INVALID); }
        uint32_t hdr[7];
        memcpy(hdr, p, 28);
        memcpy(&m->fm_b, p + 28, 4);
        m->nuf = hdr[0]; m->nif = hdr[1]; m->k = hdr[2]; m->d_user = hdr[3];
        m->th = hdr[4]; m->to = hdr[5]; m->vocab = hdr[6];
        if (m->nuf == 0 || m->nuf > 16 || m->nif * m->k != (uint32_t)pg::kDIN || !pg::fm2t_shape_ok(m->th, m->to, m->k) ||
            m->d_user == 0 || m->d_user > 4096 || m->vocab == 0) {
            pg::set_error("pg_model_load: two-tower shape unsupported (1..16 user fields; item fields x k = 128 with "
                          "k = 8, 16 or 32; towers 256-64 (k 16 / 32), 128-64 (k 8), 512-128 (k 16))");
            return fail(PG_ERR_UNSUPPORTED);
        }
        const size_t din = (size_t)m->nif * m->k;
        const size_t nf = m->nuf + m->nif;
        const size_t wfl = (size_t)m->d_user * m->th + m->th + (size_t)m->th * m->to + m->to + din * m->th +
                           m->th + (size_t)m->th * m->to + m->to;
        const size_t field_fl = nf * ((size_t)m->vocab * m->k + m->vocab);
        const size_t need = 32 + (wfl + field_fl) * 4;
        if (len != need) { pg::set_error("pg_model_load: two-tower blob is %zu bytes, expected %zu", len, need); return fail(PG_ERR_INVALID); }
        const float* uw1 = (const float*)(p + 32);
        const float* ub1 = uw1 + (size_t)m->d_user * m->th;
        const float* uw2 = ub1 + m->th;
        const float* ub2 = uw2 + (size_t)m->th * m->to;
        const float* iw1 = ub2 + m->to;
        const float* ib1 = iw1 + din * m->th;
        const float* iw2 = ib1 + m->th;
        const float* ib2 = iw2 + (size_t)m->th * m->to;
        const float* fields = ib2 + m->to;
        auto ruw1 = pg::rounded(uw1, (size_t)m->d_user * m->th, m->prec);
        auto ruw2 = pg::rounded(uw2, (size_t)m->th * m->to, m->prec);
        auto iw1p = pg::pack_weights(iw1, (uint32_t)din, m->th, m->prec);
        auto iw2p = pg::pack_weights(iw2, m->th, m->to, m->prec);
        if ((rc = pg::upload(ctx, m, ruw1.data(), ruw1.size() * 4, (void**)&m->w1u))) return fail(rc);
        if ((rc = pg::upload(ctx, m, ub1, m->th * 4, (void**)&m->b1))) return fail(rc);
        if ((rc = pg::upload(ctx, m, ruw2.data(), ruw2.size() * 4, (void**)&m->uw2))) return fail(rc);
        if ((rc = pg::upload(ctx, m, ub2, m->to * 4, (void**)&m->ub2))) return fail(rc);
        if ((rc = pg::upload(ctx, m, iw1p.data(), iw1p.size(), &m->w1p))) return fail(rc);
        if ((rc = pg::upload(ctx, m, ib1, m->th * 4, (void**)&m->c1_shared))) return fail(rc);
        if ((rc = pg::upload(ctx, m, iw2p.data(), iw2p.size(), &m->w2p))) return fail(rc);
        if (m->prec == 2) {
            m->w1p_lo = (char*)m->w1p + iw1p.size() / 2;
            m->w2p_lo = (char*)m->w2p + iw2p.size() / 2;
        }
        if ((rc = pg::upload(ctx, m, ib2, m->to * 4, (void**)&m->b2))) return fail(rc);
        if ((rc = pg::upload(ctx, m, fields, field_fl * 4, (void**)&m->fields))) return fail(rc);
        std::vector<const float*> pe(nf), pl(nf);
        for (size_t f = 0; f < nf; ++f) {
            pe[f] = m->fields + f * ((size_t)m->vocab * m->k + m->vocab);
            pl[f] = pe[f] + (size_t)m->vocab * m->k;
        }
        if ((rc = pg::upload(ctx, m, pe.data(), nf * sizeof(float*), (void**)&m->d_field_emb))) return fail(rc);
        if ((rc = pg::upload(ctx, m, pl.data(), nf * sizeof(float*), (void**)&m->d_field_lin))) return fail(rc);
    } else {
        pg::set_error("pg_model_load: unknown model kind %d", (int)kind);
        return fail(PG_ERR_INVALID);
    }
    *out = m;
    return PG_OK;
}

int pg_model_num_outputs(const pg_model* m, uint32_t* out) {
    PG_REQUIRE(m && out, "pg_model_num_outputs: NULL argument");
    *out = m->n_out;
    return PG_OK;
}

int pg_model_destroy(pg_ctx* ctx, pg_model* m) {
    PG_REQUIRE(ctx, "pg_model_destroy: ctx is NULL");
    if (!m) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    PG_HIP(hipStreamSynchronize(ctx->stream));
    for (void* a : m->allocs) PG_HIP(hipFree(a));
    delete m;
    return PG_OK;
}

int pg_rank_dnn3_dev(pg_ctx* ctx, const pg_model* m, const pg_table* t, const float* d_user_vecs,
                     const uint32_t* d_cand_rows, const uint32_t* d_req_offsets, uint32_t n_req,
                     uint32_t n_items, float* d_out_scores) {
    PG_REQUIRE(ctx && m && t && d_user_vecs && d_cand_rows && d_req_offsets && d_out_scores,
               "pg_rank_dnn3_dev: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_DNN3, "pg_rank_dnn3_dev: model is not DNN3");
    PG_REQUIRE(t->dim == m->d_item, "pg_rank_dnn3_dev: table dim %u != model d_item %u", t->dim, m->d_item);
    PG_REQUIRE(n_req <= 65535, "pg_rank_dnn3_dev: at most 65535 requests per call");
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    return pg::rank_dnn3_dev_locked(ctx, m, t, d_user_vecs, d_cand_rows, d_req_offsets, n_req, n_items,
                                    d_out_scores);
}

int pg_rank_dnn3(pg_ctx* ctx, const pg_model* m, const pg_table* t, const float* user_vecs,
                 const uint32_t* cand_rows, const uint32_t* req_offsets, uint32_t n_req,
                 float* out_scores) {
    PG_REQUIRE(ctx && m && t && req_offsets, "pg_rank_dnn3: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_DNN3, "pg_rank_dnn3: model is not DNN3");
    PG_REQUIRE(t->dim == m->d_item, "pg_rank_dnn3: table dim %u != model d_item %u", t->dim, m->d_item);
    PG_REQUIRE(n_req <= 65535, "pg_rank_dnn3: at most 65535 requests per call");
    if (n_req == 0) return PG_OK;
    PG_REQUIRE(req_offsets[0] == 0, "pg_rank_dnn3: req_offsets[0] must be 0");
    for (uint32_t r = 0; r < n_req; ++r)
        PG_REQUIRE(req_offsets[r + 1] >= req_offsets[r], "pg_rank_dnn3: req_offsets not monotone at %u", r);
    const uint32_t n_items = req_offsets[n_req];
    if (n_items == 0) return PG_OK;
    PG_REQUIRE(user_vecs && cand_rows && out_scores, "pg_rank_dnn3: NULL argument");
    for (uint32_t i = 0; i < n_items; ++i)
        PG_REQUIRE(cand_rows[i] < t->rows, "pg_rank_dnn3: candidate %u row %u outside table of %llu rows", i,
                   cand_rows[i], (unsigned long long)t->rows);
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    void* buf;
    int rc;
    const size_t ub = (size_t)n_req * m->d_user * 4, cb = (size_t)n_items * 4, ob = (size_t)(n_req + 1) * 4;
    const size_t sb = (size_t)n_items * 4 * m->n_out;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    if ((rc = pg::scratch_reserve(ctx, 5, al(ub) + al(cb) + al(ob) + al(sb), &buf))) return rc;
    char* b = (char*)buf;
    float* d_u = (float*)b;
    uint32_t* d_c = (uint32_t*)(b + al(ub));
    uint32_t* d_o = (uint32_t*)(b + al(ub) + al(cb));
    float* d_s = (float*)(b + al(ub) + al(cb) + al(ob));
    PG_HIP(hipMemcpyAsync(d_u, user_vecs, ub, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_c, cand_rows, cb, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_o, req_offsets, ob, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::rank_dnn3_dev_locked(ctx, m, t, d_u, d_c, d_o, n_req, n_items, d_s))) return rc;
    PG_HIP(hipMemcpyAsync(out_scores, d_s, sb, hipMemcpyDeviceToHost, ctx->stream));
    return pg::finish_rank_timing(ctx);
}

int pg_fm2t_user_embedding_dev(pg_ctx* ctx, const pg_model* m, const float* d_user_vecs, uint32_t n_req, float* d_out) {
    PG_REQUIRE(ctx && m && (n_req == 0 || (d_user_vecs && d_out)), "pg_fm2t_user_embedding_dev: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER, "pg_fm2t_user_embedding_dev: model is not FM_TWOTOWER");
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::fm2t_user_embedding_locked(ctx, m, d_user_vecs, n_req, d_out);
}

int pg_fm2t_user_embedding(pg_ctx* ctx, const pg_model* m, const float* user_vecs, uint32_t n_req, float* out) {
    PG_REQUIRE(ctx && m && (n_req == 0 || (user_vecs && out)), "pg_fm2t_user_embedding: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER, "pg_fm2t_user_embedding: model is not FM_TWOTOWER");
    if (n_req == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    void* buf;
    int rc;
    const size_t ub = ((size_t)n_req * m->d_user * 4 + 255) & ~(size_t)255, ob = (size_t)n_req * m->to * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, ub + ob, &buf))) return rc;
    float* d_u = (float*)buf;
    float* d_o = (float*)((char*)buf + ub);
    PG_HIP(hipMemcpyAsync(d_u, user_vecs, (size_t)n_req * m->d_user * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::fm2t_user_embedding_locked(ctx, m, d_u, n_req, d_o))) return rc;
    PG_HIP(hipMemcpyAsync(out, d_o, ob, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_rank_fm2t_dev(pg_ctx* ctx, const pg_model* m, const float* d_user_vecs,
                     const int32_t* d_user_field_ids, const int32_t* d_item_field_ids,
                     const uint32_t* d_req_offsets, uint32_t n_req, uint32_t n_items,
                     float* d_out_scores) {
    PG_REQUIRE(ctx && m && d_user_vecs && d_user_field_ids && d_item_field_ids && d_req_offsets && d_out_scores,
               "pg_rank_fm2t_dev: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER, "pg_rank_fm2t_dev: model is not FM_TWOTOWER");
    PG_REQUIRE(n_req <= 65535, "pg_rank_fm2t_dev: at most 65535 requests per call");
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::rank_fm2t_dev_locked(ctx, m, d_user_vecs, d_user_field_ids, d_item_field_ids, d_req_offsets,
                                    n_req, n_items, d_out_scores);
}

int pg_rank_fm2t_rows_dev(pg_ctx* ctx, const pg_model* m, const pg_features* fs, const int32_t* item_field_cols,
                          const float* d_user_vecs, const int32_t* d_user_field_ids, const uint32_t* d_cand_rows,
                          const uint32_t* d_req_offsets, uint32_t n_req, uint32_t n_items, float* d_out_scores) {
    PG_REQUIRE(ctx && m && fs && item_field_cols && d_user_vecs && d_user_field_ids && d_cand_rows && d_req_offsets &&
                   d_out_scores,
               "pg_rank_fm2t_rows_dev: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER, "pg_rank_fm2t_rows_dev: model is not FM_TWOTOWER");
    PG_REQUIRE(n_req <= 65535, "pg_rank_fm2t_rows_dev: at most 65535 requests per call");
    if (n_items == 0 || n_req == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::rank_fm2t_rows_dev_locked(ctx, m, fs, item_field_cols, d_user_vecs, d_user_field_ids, d_cand_rows, d_req_offsets,
                                         n_req, n_items, d_out_scores);
}

// host-buffer form of pg_rank_fm2t_rows_dev: what an IAlgorithm.Run of the EasyRec flavour passes — item ids resolved
// to rows, the per-item "context features" read from the device-resident columns instead of being boxed per request
int pg_rank_fm2t_rows(pg_ctx* ctx, const pg_model* m, const pg_features* fs, const int32_t* item_field_cols,
                      const float* user_vecs, const int32_t* user_field_ids, const uint32_t* cand_rows,
                      const uint32_t* req_offsets, uint32_t n_req, float* out_scores) {
    PG_REQUIRE(ctx && m && fs && item_field_cols && req_offsets, "pg_rank_fm2t_rows: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER, "pg_rank_fm2t_rows: model is not FM_TWOTOWER");
    PG_REQUIRE(n_req <= 65535, "pg_rank_fm2t_rows: at most 65535 requests per call");
    if (n_req == 0) return PG_OK;
    PG_REQUIRE(req_offsets[0] == 0, "pg_rank_fm2t_rows: req_offsets[0] must be 0");
    for (uint32_t r = 0; r < n_req; ++r)
        PG_REQUIRE(req_offsets[r + 1] >= req_offsets[r], "pg_rank_fm2t_rows: req_offsets not monotone at %u", r);
    const uint32_t n_items = req_offsets[n_req];
    if (n_items == 0) return PG_OK;
    PG_REQUIRE(user_vecs && user_field_ids && cand_rows && out_scores, "pg_rank_fm2t_rows: NULL argument");
    for (size_t i = 0; i < (size_t)n_req * m->nuf; ++i)
        PG_REQUIRE(user_field_ids[i] >= 0 && (uint32_t)user_field_ids[i] < m->vocab,
                   "pg_rank_fm2t_rows: user field id %d outside vocab %u", user_field_ids[i], m->vocab);
    std::lock_guard<std::mutex> g(ctx->mu);
    void* buf;
    int rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t ub = (size_t)n_req * m->d_user * 4, ufb = (size_t)n_req * m->nuf * 4, cb = (size_t)n_items * 4;
    const size_t ifb = (size_t)n_items * m->nif * 4, ob = (size_t)(n_req + 1) * 4, sb = (size_t)n_items * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, al(ub) + al(ufb) + al(cb) + al(ifb) + al(ob) + al(sb), &buf))) return rc;
    char* b = (char*)buf;
    float* d_u = (float*)b; b += al(ub);
    int32_t* d_uf = (int32_t*)b; b += al(ufb);
    uint32_t* d_c = (uint32_t*)b; b += al(cb);
    int32_t* d_if = (int32_t*)b; b += al(ifb);
    uint32_t* d_o = (uint32_t*)b; b += al(ob);
    float* d_s = (float*)b;
    PG_HIP(hipMemcpyAsync(d_u, user_vecs, ub, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_uf, user_field_ids, ufb, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_c, cand_rows, cb, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_o, req_offsets, ob, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::features_gather_i32_locked(ctx, fs, item_field_cols, m->nif, d_c, n_items, d_if, "pg_rank_fm2t_rows"))) return rc;
    if ((rc = pg::rank_fm2t_dev_locked(ctx, m, d_u, d_uf, d_if, d_o, n_req, n_items, d_s))) return rc;
    PG_HIP(hipMemcpyAsync(out_scores, d_s, sb, hipMemcpyDeviceToHost, ctx->stream));
    return pg::finish_rank_timing(ctx);
}

int pg_rank_fm2t(pg_ctx* ctx, const pg_model* m, const float* user_vecs, const int32_t* user_field_ids,
                 const int32_t* item_field_ids, const uint32_t* req_offsets, uint32_t n_req,
                 float* out_scores) {
    PG_REQUIRE(ctx && m && req_offsets, "pg_rank_fm2t: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER, "pg_rank_fm2t: model is not FM_TWOTOWER");
    PG_REQUIRE(n_req <= 65535, "pg_rank_fm2t: at most 65535 requests per call");
    if (n_req == 0) return PG_OK;
    PG_REQUIRE(req_offsets[0] == 0, "pg_rank_fm2t: req_offsets[0] must be 0");
    for (uint32_t r = 0; r < n_req; ++r)
        PG_REQUIRE(req_offsets[r + 1] >= req_offsets[r], "pg_rank_fm2t: req_offsets not monotone at %u", r);
    const uint32_t n_items = req_offsets[n_req];
    if (n_items == 0) return PG_OK;
    PG_REQUIRE(user_vecs && user_field_ids && item_field_ids && out_scores, "pg_rank_fm2t: NULL argument");
    for (size_t i = 0; i < (size_t)n_req * m->nuf; ++i)
        PG_REQUIRE(user_field_ids[i] >= 0 && (uint32_t)user_field_ids[i] < m->vocab,
                   "pg_rank_fm2t: user field id %d outside vocab %u", user_field_ids[i], m->vocab);
    for (size_t i = 0; i < (size_t)n_items * m->nif; ++i)
        PG_REQUIRE(item_field_ids[i] >= 0 && (uint32_t)item_field_ids[i] < m->vocab,
                   "pg_rank_fm2t: item field id %d outside vocab %u", item_field_ids[i], m->vocab);
    std::lock_guard<std::mutex> g(ctx->mu);
    void* buf;
    int rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t ub = (size_t)n_req * m->d_user * 4, ufb = (size_t)n_req * m->nuf * 4;
    const size_t ifb = (size_t)n_items * m->nif * 4, ob = (size_t)(n_req + 1) * 4, sb = (size_t)n_items * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, al(ub) + al(ufb) + al(ifb) + al(ob) + al(sb), &buf))) return rc;
    char* b = (char*)buf;
    float* d_u = (float*)b; b += al(ub);
    int32_t* d_uf = (int32_t*)b; b += al(ufb);
    int32_t* d_if = (int32_t*)b; b += al(ifb);
    uint32_t* d_o = (uint32_t*)b; b += al(ob);
    float* d_s = (float*)b;
    PG_HIP(hipMemcpyAsync(d_u, user_vecs, ub, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_uf, user_field_ids, ufb, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_if, item_field_ids, ifb, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_o, req_offsets, ob, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::rank_fm2t_dev_locked(ctx, m, d_u, d_uf, d_if, d_o, n_req, n_items, d_s))) return rc;
    PG_HIP(hipMemcpyAsync(out_scores, d_s, sb, hipMemcpyDeviceToHost, ctx->stream));
    return pg::finish_rank_timing(ctx);
}

int pg_fm2t_item_rows_build(pg_ctx* ctx, const pg_model* m, const pg_features* fs, const int32_t* item_field_cols,
                            pg_item_rows** out) {
    PG_REQUIRE(ctx && m && fs && item_field_cols && out, "pg_fm2t_item_rows_build: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER, "pg_fm2t_item_rows_build: model is not FM_TWOTOWER");
    PG_REQUIRE(m->nif <= 16 && m->nif * m->k == (uint32_t)pg::kDIN, "pg_fm2t_item_rows_build: unsupported item side (%u fields x %u)", m->nif, m->k);
    PG_REQUIRE(fs->rows < 0xFFFFFFFEull, "pg_fm2t_item_rows_build: too many rows");
    std::lock_guard<std::mutex> g(ctx->mu);
    PG_HIP(hipSetDevice(ctx->device));
    pg_item_rows* ir = new pg_item_rows();
    ir->m = m;
    ir->fs = fs;
    ir->rows = fs->rows;
    for (uint32_t f = 0; f < m->nif; ++f) ir->cols[f] = item_field_cols[f];
    const size_t bytes = (size_t)(fs->rows + 1) * pg::kItemRowFloats * 4;
    const hipError_t e = hipMalloc((void**)&ir->d, bytes);
    if (e != hipSuccess) {
        pg::set_error("pg_fm2t_item_rows_build: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        delete ir;
        return PG_ERR_NOMEM;
    }
    int rc;
    if ((rc = pg::item_rows_fill_locked(ctx, ir, 0, fs->rows + 1))) {       // + the defaults' record
        hipFree(ir->d);
        delete ir;
        return rc;
    }
    PG_HIP(hipStreamSynchronize(ctx->stream));
    *out = ir;
    return PG_OK;
}

int pg_fm2t_item_rows_update(pg_ctx* ctx, pg_item_rows* ir, uint64_t row0, uint64_t nrows) {
    PG_REQUIRE(ctx && ir, "pg_fm2t_item_rows_update: NULL argument");
    PG_REQUIRE(row0 + nrows <= ir->rows, "pg_fm2t_item_rows_update: rows %llu..%llu outside the store", (unsigned long long)row0,
               (unsigned long long)(row0 + nrows));
    std::lock_guard<std::mutex> g(ctx->mu);
    int rc;
    if ((rc = pg::item_rows_fill_locked(ctx, ir, row0, nrows))) return rc;
    if ((rc = pg::item_rows_fill_locked(ctx, ir, ir->rows, 1))) return rc;     // the defaults may have changed with the column
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_fm2t_item_rows_destroy(pg_ctx* ctx, pg_item_rows* ir) {
    PG_REQUIRE(ctx, "pg_fm2t_item_rows_destroy: NULL context");
    if (!ir) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    PG_HIP(hipStreamSynchronize(ctx->stream));
    if (ir->d) PG_HIP(hipFree(ir->d));
    delete ir;
    return PG_OK;
}

int pg_rank_fm2t_irows_dev(pg_ctx* ctx, const pg_model* m, const pg_item_rows* ir, const float* d_user_vecs,
                           const int32_t* d_user_field_ids, const uint32_t* d_cand_rows, const uint32_t* d_req_offsets,
                           uint32_t n_req, uint32_t n_items, float* d_out_scores) {
    PG_REQUIRE(ctx && m && ir && d_user_vecs && d_user_field_ids && d_cand_rows && d_req_offsets && d_out_scores,
               "pg_rank_fm2t_irows_dev: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER && ir->m == m, "pg_rank_fm2t_irows_dev: the item records belong to another model");
    PG_REQUIRE(n_req <= 65535, "pg_rank_fm2t_irows_dev: at most 65535 requests per call");
    if (n_items == 0 || n_req == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::rank_fm2t_irows_dev_locked(ctx, m, ir, d_user_vecs, d_user_field_ids, d_cand_rows, d_req_offsets, n_req, n_items, d_out_scores);
}

int pg_rank_fm2t_irows(pg_ctx* ctx, const pg_model* m, const pg_item_rows* ir, const float* user_vecs,
                       const int32_t* user_field_ids, const uint32_t* cand_rows, const uint32_t* req_offsets, uint32_t n_req,
                       float* out_scores) {
    PG_REQUIRE(ctx && m && ir && req_offsets, "pg_rank_fm2t_irows: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER && ir->m == m, "pg_rank_fm2t_irows: the item records belong to another model");
    PG_REQUIRE(n_req <= 65535, "pg_rank_fm2t_irows: at most 65535 requests per call");
    if (n_req == 0) return PG_OK;
    PG_REQUIRE(req_offsets[0] == 0, "pg_rank_fm2t_irows: req_offsets[0] must be 0");
    for (uint32_t r = 0; r < n_req; ++r)
        PG_REQUIRE(req_offsets[r + 1] >= req_offsets[r], "pg_rank_fm2t_irows: req_offsets not monotone at %u", r);
    const uint32_t n_items = req_offsets[n_req];
    if (n_items == 0) return PG_OK;
    PG_REQUIRE(user_vecs && user_field_ids && cand_rows && out_scores, "pg_rank_fm2t_irows: NULL argument");
    for (size_t i = 0; i < (size_t)n_req * m->nuf; ++i)
        PG_REQUIRE(user_field_ids[i] >= 0 && (uint32_t)user_field_ids[i] < m->vocab,
                   "pg_rank_fm2t_irows: user field id %d outside vocab %u", user_field_ids[i], m->vocab);
    std::lock_guard<std::mutex> g(ctx->mu);
    void* buf;
    int rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t ub = (size_t)n_req * m->d_user * 4, ufb = (size_t)n_req * m->nuf * 4, cb = (size_t)n_items * 4;
    const size_t ob = (size_t)(n_req + 1) * 4, sb = (size_t)n_items * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, al(ub) + al(ufb) + al(cb) + al(ob) + al(sb), &buf))) return rc;
    char* b = (char*)buf;
    float* d_u = (float*)b; b += al(ub);
    int32_t* d_uf = (int32_t*)b; b += al(ufb);
    uint32_t* d_c = (uint32_t*)b; b += al(cb);
    uint32_t* d_o = (uint32_t*)b; b += al(ob);
    float* d_s = (float*)b;
    PG_HIP(hipMemcpyAsync(d_u, user_vecs, ub, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_uf, user_field_ids, ufb, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_c, cand_rows, cb, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_o, req_offsets, ob, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::rank_fm2t_irows_dev_locked(ctx, m, ir, d_u, d_uf, d_c, d_o, n_req, n_items, d_s))) return rc;
    PG_HIP(hipMemcpyAsync(out_scores, d_s, sb, hipMemcpyDeviceToHost, ctx->stream));
    return pg::finish_rank_timing(ctx);
}

}  // extern "C"
