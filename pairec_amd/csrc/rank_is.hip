// rank_is.hip — FM + two-tower rank over materialised item records with EVERY wave a whole pipeline (no producer /
// consumer split, no workgroup barrier in the loop, no activation tile in LDS).
#include "rank_mlp.hpp"

#include <cstdlib>

namespace pg {

// ---------------------------------------------------------------------------------------------
// fm2t_isw_kernel: the same model, shape and arithmetic as fm2t_irs_kernel (rank_ir.hip; algorithm/eas/fm_request.go:29-79,
// service/rank/rank_service.go:264-289) — 8 item fields x 16, item tower 128 -> 256 -> 64, bf16, 640-B item records.
//
// fm2t_irs_kernel keeps the towers in the registers of four consumer waves and feeds them through LDS tiles from four
// producer waves: two barrier-coupled halves per 64-item tile, each as long as its slowest wave, 5 500 cycles per tile
// against an HBM floor of 3 400.  Here the towers' 96 KB of MFMA fragments live in LDS, read-only, and a
// wave owns a 32-item tile from its records to its scores:
//   * gather: four adjacent lanes per record, two passes of sixteen records, straight into registers (eight 16-B quads +
//     the linear quad per lane and pass); the FM chains run in the lane, in the specification's order, exactly as
//     fm2t_irs_kernel's conversion does;
//   * the X operand never exists as a tile: layer 1 is computed in swapped form (C^T = W^T X^T), whose B operand wants
//     lane (item, h) to hold dims 16 f + 8 h .. + 7 of its item — the packed quads 2 h, 2 h + 1 of field f, which sit in
//     lanes (item, 2 h), (item, 2 h + 1) of the gather layout: two `ds_bpermute` per dword (one per pass) and a select;
//   * H1 never leaves the registers either: a 32 x 32 block of layer 1's output has lane (item, h) holding hidden columns
//     8 g + 4 h + r, and layer 2's B operand wants 16 u + 8 h + 0..7 — after relu and packing, ONE `v_permlane32_swap` per
//     register pair turns (P_0, P_1) into the fragment of k-step 2 nb and (P_2, P_3) into that of 2 nb + 1;
//   * the head's two 32-column chains cross the lane halves every four columns the same way (sixteen hand-overs);
//   * per hidden block: 8 layer-1 MFMAs (one chain, k ascending, from the bias) and 4 layer-2 MFMAs (two output blocks,
//     k-steps 2 nb, 2 nb + 1) — the accumulation orders of both layers are fm2t_irs_kernel's, hence mlp_kernel's, hence
//     the per-field path's: scores are bit-identical (test_fm2t_materialised_item_records_are_bit_identical).
// Eight waves per CU (two per SIMD, 246 registers) run out of phase by themselves: one's gather latency and VALU phases
// lie under the others' MFMAs.  The records of a wave's NEXT tile are requested before its towers run (72 registers in
// flight under the MFMAs), its candidate rows a trip earlier, its descriptor a trip before that; the request's FM prefix
// and user-tower output sit in a wave-private LDS cache refilled when the request changes (~150 tiles).
// Measured (256 x 5 000 random candidates of a 20 M-item catalogue, one MI355X): rank stage 0.226-0.230 ms against
// fm2t_irs_kernel's 0.246-0.262 on the same box, 0.219-0.227 in bench.py's leg (0.241); with every candidate = row 0 0.15 ms
// (0.19).  The gather + FM sums alone (-DPG_ISW_GATHER_ONLY) take 0.171 ms = 5.3 TB/s of record lines: the towers' LDS
// and MFMA traffic costs the memory side a quarter of that (waves wait 55 % of their lifetime for their records, all
// 2 048 of them with 18 KB in flight).  Tried: twelve waves (168 registers: 99 spilled dwords with the prefetch, 0.28 ms
// without it), whole-line fetches with eight lanes per record (probe: -2 % on the gather alone, nothing under the towers —
// unlike fm2t_irs_kernel's LDS-DMAs, where it gave 16 %), non-temporal loads (nothing).
// ---------------------------------------------------------------------------------------------
constexpr int kIsTH = 256, kIsTO = 64;
constexpr size_t kIsW1 = (size_t)kDIN * kIsTH * 2;            // 64 KiB of layer-1 fragments [n-block][k-step][lane]
constexpr size_t kIsW2 = (size_t)kIsTH * kIsTO * 2;           // 32 KiB of layer-2 fragments
constexpr int kIsRq = 132;                                    // floats of a wave's request cache: s[32] | q[32] | lin | pad | tower output [64] at 68
constexpr size_t is_lds_bytes(int waves) { return kIsW1 + kIsW2 + kIsTH * 4 + kIsTO * 4 + (size_t)waves * kIsRq * 4; }

__device__ __forceinline__ float is_lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float is_lane_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float is_from_lane_minus1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false));
}
// max(v, +0) with NaN -> 0, -0 -> +0 (mlp_kernel's ternary; compiler-visible: an asm v_max reading a fresh MFMA result is
// not seen by the hazard recognizer)
__device__ __forceinline__ float is_relu(float v) { return v > 0.0f ? v : 0.0f; }
__device__ __forceinline__ uint32_t is_pack(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
}
// v_permlane32_swap_b32 x, y: lanes 32..63 of x and lanes 0..31 of y change places — x = {x.lo, y.lo}, y = {x.hi, y.hi}.
// As inline asm: hipcc (ROCm 7.2) loses track of which result is which when the builtin sits in a dependent chain
// (scripts/micro/permlane_swap.hip: three round trips come back as one); the s_nops are the wait states between a VALU
// write of an operand and the swap, which the hazard recognizer cannot add inside an asm.
__device__ __forceinline__ void is_swap32(uint32_t& x, uint32_t& y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(x), "+v"(y));
}
// lanes 32..63 receive the value lanes 0..31 hold / lanes 0..31 receive what lanes 32..63 hold
__device__ __forceinline__ float is_to_upper(float v) {
    uint32_t z = 0u, x = __builtin_bit_cast(uint32_t, v);
    is_swap32(z, x);
    return __builtin_bit_cast(float, z);
}
__device__ __forceinline__ float is_to_lower(float v) {
    uint32_t x = __builtin_bit_cast(uint32_t, v), z = 0u;
    is_swap32(x, z);
    return __builtin_bit_cast(float, z);
}

#ifndef PG_ISW_WAVES
#define PG_ISW_WAVES 8
#endif
constexpr int kIsWaves = PG_ISW_WAVES;                         // per CU (one workgroup)

__global__ __launch_bounds__(64 * kIsWaves, 1) void fm2t_isw_kernel(MlpArgs a) {
    constexpr int KS1 = kDIN / 16, KS2 = kIsTH / 16, NB1 = kIsTH / 32;     // 8 k-steps / 16 k-steps / 8 hidden blocks
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const W1S = smem;
    char* const W2S = smem + kIsW1;
    float* const c1s = reinterpret_cast<float*>(W2S + kIsW2);
    float* const b2s = c1s + kIsTH;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const rq = b2s + kIsTO + wave * kIsRq;             // this wave's request cache (filled when its tile's request changes)
    const uint32_t n_tiles = *a.n_tiles;
    const uint32_t t_begin = (uint32_t)(((uint64_t)n_tiles * blockIdx.x) / gridDim.x);
    const uint32_t t_end = (uint32_t)(((uint64_t)n_tiles * (blockIdx.x + 1)) / gridDim.x);

    // ---- software pipeline over this wave's tiles t, t + W, ..: the records of tile t + W are requested before tile t's
    // towers run (their 72 registers stay in flight under the MFMAs), its candidate rows a trip earlier, its descriptor a
    // trip before that — nothing the loop waits for was requested in the same trip
    struct Desc {
        uint32_t req, item0, cnt;
    };
    auto load_desc = [&](uint32_t t) {
        const uint32_t tc = t < t_end ? t : (t_end ? t_end - 1 : 0u);   // (past the range: a valid entry, whose loads are never used)
        return Desc{a.tile_req[tc], a.tile_item0[tc], a.tile_cnt[tc]};      // (uniform: scalar loads)
    };
    auto load_rows = [&](const Desc& d, uint32_t (&rows)[2]) {
        uint32_t l_ = (uint32_t)lane;
        asm volatile("" : "+v"(l_));
        const uint32_t last = d.cnt ? d.cnt - 1 : 0u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t it = 16u * p + (l_ >> 2);
            rows[p] = a.cand_rows[d.item0 + (it < last ? it : last)];
        }
    };
    float4 e[2][8], lq[2];
    // lane (r, j) = quad j of every field of records r (pass 0) and 16 + r (pass 1)
    auto issue_gather = [&](const uint32_t (&rows)[2]) {
        uint32_t l_ = (uint32_t)lane;
        asm volatile("" : "+v"(l_));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t row = rows[p] < a.irow_count ? rows[p] : a.irow_count;     // (outside the store: the defaults' record)
            const char* const rec = reinterpret_cast<const char*>(a.irows) + (size_t)row * (kItemRowFloats * 4) + (l_ & 3) * 16;
#pragma unroll
            for (int f = 0; f < 8; ++f) e[p][f] = *reinterpret_cast<const float4*>(rec + f * 64);
            lq[p] = *reinterpret_cast<const float4*>(rec + kDIN * 4);       // (lanes 0 / 1 of the record: the linear weights)
        }
    };
    uint32_t t = t_begin + (uint32_t)wave;
    Desc d0 = load_desc(t), d1 = load_desc(t + kIsWaves), d2 = load_desc(t + 2 * kIsWaves);
    uint32_t rows1[2];
    {
        uint32_t rows0[2];
        load_rows(d0, rows0);
        load_rows(d1, rows1);
        if (t < t_end) issue_gather(rows0);
    }
    // ---- the towers -> LDS, once — behind the first tile's record loads, which fly meanwhile
    {
        const uint4* const g1 = reinterpret_cast<const uint4*>(a.w1p);
        const uint4* const g2 = reinterpret_cast<const uint4*>(a.w2p);
        uint4* const s1 = reinterpret_cast<uint4*>(W1S);
        uint4* const s2 = reinterpret_cast<uint4*>(W2S);
        for (uint32_t i = tid; i < kIsW1 / 16; i += 64 * kIsWaves) s1[i] = g1[i];
        for (uint32_t i = tid; i < kIsW2 / 16; i += 64 * kIsWaves) s2[i] = g2[i];
        if (tid < kIsTH) c1s[tid] = a.c1[tid];
        if (tid < kIsTO) b2s[tid] = a.b2[tid];
    }
    __syncthreads();
    uint32_t cached_req = 0xFFFFFFFFu;
    for (; t < t_end; t += kIsWaves) {
        uint32_t l_ = (uint32_t)lane;
        asm volatile("" : "+v"(l_));                           // (per-lane indices re-derived per tile: hoisted, they are spilled)
        const uint32_t req = d0.req, item0 = d0.item0, cnt = d0.cnt;
        const uint32_t gj = l_ & 3;
        // the request's FM prefix and user-tower output: through the wave's LDS cache (a request is ~150 tiles long; read
        // from global per tile, these eleven loads were the exposed latencies of the trip)
        if (req != cached_req) {
            const float* const fu = a.fm_user + (size_t)req * kFmUserStride;
            rq[l_] = fu[1 + l_];                               // s[0..31] | q[0..31]
            if (l_ == 0) rq[64] = fu[0];
            rq[68 + l_] = a.w3[(size_t)req * a.w3_stride + l_];
            cached_req = req;
        }
        const float4 s4 = *reinterpret_cast<const float4*>(rq + 4 * gj);
        const float4 q4 = *reinterpret_cast<const float4*>(rq + kFmMaxK + 4 * gj);
        const float linu = rq[64];

        // ---- FM terms and the packed quads (fm2t_irs_kernel's conversion, per pass)
        uint32_t qp[2][8][2];
        float fmt[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float sc[4] = {s4.x, s4.y, s4.z, s4.w}, qc[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (int f = 0; f < 8; ++f) {                      // user prefix first, fields ascending
                const float xv[4] = {e[p][f].x, e[p][f].y, e[p][f].z, e[p][f].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    asm("v_add_f32 %0, %1, %0" : "+v"(sc[c]) : "v"(xv[c]));
                    asm("v_fma_f32 %0, %1, %1, %0" : "+v"(qc[c]) : "v"(xv[c]));
                }
                qp[p][f][0] = is_pack(xv[0], xv[1]);
                qp[p][f][1] = is_pack(xv[2], xv[3]);
            }
            float s_[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) s_[c] = __fmaf_rn(sc[c], sc[c], -qc[c]);
            float cross = (s_[0] + s_[1]) + (s_[2] + s_[3]);   // tree levels 1, 2 (columns of one quad)
            cross = cross + is_lane_xor1(cross);               // level 3
            cross = cross + is_lane_xor2(cross);               // level 4
            // linear term: prefix + the eight weights one by one — lane 0 adds 0..3, lane 1 (from lane 0's sum) 4..7
            float lin = linu;
            lin = lin + lq[p].x; lin = lin + lq[p].y; lin = lin + lq[p].z; lin = lin + lq[p].w;
            float lin1 = is_from_lane_minus1(lin);
            lin1 = lin1 + lq[p].x; lin1 = lin1 + lq[p].y; lin1 = lin1 + lq[p].z; lin1 = lin1 + lq[p].w;
            fmt[p] = lin1 + 0.5f * cross;                      // (valid in the record's lane 1)
        }
        // ---- the B operand of layer 1: lane (item, h) <- quads 2 h, 2 h + 1 of its record's fields
        const uint32_t bi = l_ & 31, bh = l_ >> 5;
        const int src_lo = (int)(((bi & 15) * 4 + 2 * bh) * 4), src_hi = src_lo + 4;      // (byte addresses of the source lanes)
        const bool second = bi >= 16;
        bf16x8 xb[KS1];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            uint32_t w[4];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int lo0 = __builtin_amdgcn_ds_bpermute(src_lo, (int)qp[0][f][d]), lo1 = __builtin_amdgcn_ds_bpermute(src_lo, (int)qp[1][f][d]);
                const int hi0 = __builtin_amdgcn_ds_bpermute(src_hi, (int)qp[0][f][d]), hi1 = __builtin_amdgcn_ds_bpermute(src_hi, (int)qp[1][f][d]);
                w[d] = (uint32_t)(second ? lo1 : lo0);
                w[2 + d] = (uint32_t)(second ? hi1 : hi0);
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            xb[f] = __builtin_bit_cast(bf16x8, u32x4{w[0], w[1], w[2], w[3]});
        }
        // the item's FM term, into its lane of the lower half (chain 0 of the head starts from it)
        const int fsrc = (int)(((bi & 15) * 4 + 1) * 4);
        const float f0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(fsrc, __builtin_bit_cast(int, fmt[0])));
        const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(fsrc, __builtin_bit_cast(int, fmt[1])));
        const float fm_term = second ? f1 : f0;

        // ---- the next tiles' loads (e / lq are free again)
        const Desc d3 = load_desc(t + 3 * kIsWaves);
        if (t + kIsWaves < t_end) issue_gather(rows1);
        load_rows(d2, rows1);                                  // (tile t + 2 W's rows: requested behind the gather that used the old ones)

#ifdef PG_ISW_GATHER_ONLY
        {                                                      // (developer experiment: the gather + FM + permutes alone)
            float v = fm_term;
#pragma unroll
            for (int f = 0; f < 8; ++f)
#pragma unroll
                for (int i = 0; i < 8; ++i) v += (float)xb[f][i];
            if (bh == 1 && bi < cnt) a.out[item0 + bi] = v;
            d0 = d1;
            d1 = d2;
            d2 = d3;
            continue;
        }
#endif
        // ---- the towers: per hidden block 8 + 4 MFMAs
        f32x16 acc2[2];
#pragma unroll
        for (int nb2 = 0; nb2 < 2; ++nb2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(b2s + nb2 * 32 + 4 * bh + 8 * g);
                acc2[nb2][4 * g + 0] = bv.x; acc2[nb2][4 * g + 1] = bv.y; acc2[nb2][4 * g + 2] = bv.z; acc2[nb2][4 * g + 3] = bv.w;
            }
#pragma unroll
        for (int nb = 0; nb < NB1; ++nb) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 cv = *reinterpret_cast<const float4*>(c1s + nb * 32 + 4 * bh + 8 * g);
                acc[4 * g + 0] = cv.x; acc[4 * g + 1] = cv.y; acc[4 * g + 2] = cv.z; acc[4 * g + 3] = cv.w;
            }
            bf16x8 wf[KS1];
#pragma unroll
#ifdef PG_ISW_NO_LDSW                                           // (developer experiment: the towers without their LDS traffic — wrong results)
            for (int ks = 0; ks < KS1; ++ks) wf[ks] = xb[(ks + nb) & 7];
#else
            for (int ks = 0; ks < KS1; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(W1S + (size_t)(nb * KS1 + ks) * 1024 + l_ * 16);
#endif
            bf16x8 w2f[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int nb2 = 0; nb2 < 2; ++nb2)
#ifdef PG_ISW_NO_LDSW
                    w2f[u][nb2] = xb[(u * 2 + nb2 + nb) & 7];
#else
                    w2f[u][nb2] = *reinterpret_cast<const bf16x8*>(W2S + (size_t)(nb2 * KS2 + 2 * nb + u) * 1024 + l_ * 16);
#endif
#pragma unroll
#ifdef PG_ISW_NO_MFMA                                           // (developer experiment: the LDS traffic without the MFMAs — wrong results)
            for (int ks = 0; ks < KS1; ++ks) acc[ks] += (float)wf[ks][0] * (float)xb[ks][1];
#else
            for (int ks = 0; ks < KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xb[ks], acc, 0, 0, 0);
#endif
            // relu -> bf16 -> the two k-steps' B fragments of layer 2
            uint32_t pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pk[g][0] = is_pack(is_relu(acc[4 * g + 0]), is_relu(acc[4 * g + 1]));
                pk[g][1] = is_pack(is_relu(acc[4 * g + 2]), is_relu(acc[4 * g + 3]));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                is_swap32(pk[2 * u][0], pk[2 * u + 1][0]);      // -> {own / partner's columns 0..3 | 8..11}, {4..7 | 12..15} of the k-step
                is_swap32(pk[2 * u][1], pk[2 * u + 1][1]);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const bf16x8 hb = __builtin_bit_cast(bf16x8, u32x4{pk[2 * u][0], pk[2 * u][1], pk[2 * u + 1][0], pk[2 * u + 1][1]});
#pragma unroll
#ifdef PG_ISW_NO_MFMA
                for (int nb2 = 0; nb2 < 2; ++nb2) acc2[nb2][u] += (float)w2f[u][nb2][0] * (float)hb[1];
#else
                for (int nb2 = 0; nb2 < 2; ++nb2) acc2[nb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[u][nb2], hb, acc2[nb2], 0, 0, 0);
#endif
            }
        }
        // ---- head: chain 0 over output columns 0..31 (from the FM term), chain 1 over 32..63 (from 0), both ascending; a
        // lane half owns columns 8 g + 4 h + 0..3 of a block, so the chains change halves every four columns
        const float* const w3 = rq + 68;
        float c0 = fm_term, c1 = 0.0f;                          // (meaningful in the half whose turn it is)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 wa = *reinterpret_cast<const float4*>(w3 + 8 * g + 4 * bh);
            const float4 wb = *reinterpret_cast<const float4*>(w3 + 32 + 8 * g + 4 * bh);
            if (g) {                                           // from the upper half's columns 8 g - 4 .. back to the lower half
                c0 = is_to_lower(c0);
                c1 = is_to_lower(c1);
            }
            // (both halves execute every step; only the owner's result is carried on)
            float a0 = c0, a1 = c1;
            a0 = __fmaf_rn(acc2[0][4 * g + 0], wa.x, a0); a0 = __fmaf_rn(acc2[0][4 * g + 1], wa.y, a0);
            a0 = __fmaf_rn(acc2[0][4 * g + 2], wa.z, a0); a0 = __fmaf_rn(acc2[0][4 * g + 3], wa.w, a0);
            a1 = __fmaf_rn(acc2[1][4 * g + 0], wb.x, a1); a1 = __fmaf_rn(acc2[1][4 * g + 1], wb.y, a1);
            a1 = __fmaf_rn(acc2[1][4 * g + 2], wb.z, a1); a1 = __fmaf_rn(acc2[1][4 * g + 3], wb.w, a1);
            // lower half has done columns 8 g .. 8 g + 3; the upper half continues from there with 8 g + 4 .. 8 g + 7
            float b0 = is_to_upper(a0), b1 = is_to_upper(a1);
            b0 = __fmaf_rn(acc2[0][4 * g + 0], wa.x, b0); b0 = __fmaf_rn(acc2[0][4 * g + 1], wa.y, b0);
            b0 = __fmaf_rn(acc2[0][4 * g + 2], wa.z, b0); b0 = __fmaf_rn(acc2[0][4 * g + 3], wa.w, b0);
            b1 = __fmaf_rn(acc2[1][4 * g + 0], wb.x, b1); b1 = __fmaf_rn(acc2[1][4 * g + 1], wb.y, b1);
            b1 = __fmaf_rn(acc2[1][4 * g + 2], wb.z, b1); b1 = __fmaf_rn(acc2[1][4 * g + 3], wb.w, b1);
            c0 = b0;                                           // (valid in the upper half)
            c1 = b1;
        }
        const float z = c0 + c1;
#ifdef PG_ISW_DEBUG
        const int dbg = (int)a.b3;
        if (dbg) {
            float v = 0.0f;
            if (dbg == 1) v = fm_term;
            if (dbg == 2 || dbg == 3) {
#pragma unroll
                for (int f = 0; f < 8; ++f)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v += (float)xb[f][i];
            }
            if (dbg == 4 || dbg == 5) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v += acc2[0][i];
            }
            if (dbg == 6) v = c0;
            if (dbg == 7) v = c1;
            const bool upper = dbg == 3 || dbg == 5 || dbg == 6 || dbg == 7;
            if ((bh == 1) == upper && bi < cnt) a.out[item0 + bi] = v;
            d0 = d1;
            d1 = d2;
            d2 = d3;
            continue;
        }
#endif
        if (bh == 1 && bi < cnt) a.out[item0 + bi] = 1.0f / (1.0f + expf(-z));
        d0 = d1;
        d1 = d2;
        d2 = d3;
    }
}

bool fm2t_isw_shape(uint32_t th, uint32_t to, uint32_t k, uint32_t nif, int prec) { return prec == 1 && th == 256 && to == 64 && k == 16 && nif == 8; }

int launch_fm2t_isw(pg_ctx* ctx, const MlpArgs& a) {
    constexpr size_t lds = is_lds_bytes(kIsWaves);
    int rc;
    if ((rc = ensure_dyn_lds(ctx, (const void*)fm2t_isw_kernel, lds))) return rc;
#ifdef PG_ISW_DEBUG
    MlpArgs b = a;
    b.b3 = getenv("PG_ISW_DEBUG_MODE") ? (float)atoi(getenv("PG_ISW_DEBUG_MODE")) : 0.0f;
    fm2t_isw_kernel<<<ctx->num_cus, 64 * kIsWaves, lds, ctx->stream>>>(b);
    return PG_OK;
#endif
    fm2t_isw_kernel<<<ctx->num_cus, 64 * kIsWaves, lds, ctx->stream>>>(a);
    return PG_OK;
}

}  // namespace pg
