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
#include "pipeline.hpp"
#include "bitonic_reg.hpp"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#ifndef DPP_ABL
#define DPP_ABL 0
#endif
namespace pg {

// Feature rows, one thread per candidate of one request (blockIdx.y = request), following KernelMatrix
// (dpp_sort.go:408-447):
//   table path (has_table): e = item embedding, L2-normalised when normalize (loadEmbeddingCache :235-236:
//     floats.Norm / floats.Scale(1/norm)); with hook embeddings (RegisterEmbeddingHook, :362,413) the row is
//     [hook ‖ e] re-normalised jointly (:419-421); then "append 1, scale 1/√2" (:428-430) — EnsurePositiveSim is
//     not consulted on this path in the reference;
//   hook-only path: c = hook, normalised when normalize (:437-440); ensure_pos → [c,1]/√2, else [c,0] unscaled (:441-446).
// r_i = exp(alpha * relevance_i) (:431).  emb32 rows are fp32 [R][n][d]; hook rows fp64 [R][n][hook_dim].
struct DppPrep {
    const float* emb32;
    const double* hook;
    const double* rel;
    uint32_t n, d, hook_dim;
    double alpha;
    int normalize, ensure_pos, has_table;
    double* F;       // [R][n][d1]
    double* r;       // [R][n]
};
__global__ void dpp_prepare_kernel(DppPrep a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = blockIdx.y;
    if (i >= a.n) return;
    const uint32_t dt = a.has_table ? a.d : 0u;
    const uint32_t w = a.hook_dim + dt;            // feature width before the appended constant
    const uint32_t d1 = w + 1;
    const size_t item = (size_t)q * a.n + i;
    double* f = a.F + item * d1;
    const double isq2 = 0.70710678118654757;      // Go constant 1/math.Sqrt2
    // stage the row in F itself: [hook ‖ e]
    for (uint32_t k = 0; k < a.hook_dim; ++k) f[k] = a.hook[item * a.hook_dim + k];
    if (a.has_table) {
        const float* x = a.emb32 + item * a.d;
        double inv = 1.0;
        if (a.normalize) {
            double ss = 0.0;
            for (uint32_t k = 0; k < a.d; ++k) {
                const double v = (double)x[k];
                ss = fma(v, v, ss);
            }
            inv = 1.0 / sqrt(ss);
        }
        for (uint32_t k = 0; k < a.d; ++k) {
            const double v = (double)x[k];
            f[a.hook_dim + k] = a.normalize ? inv * v : v;
        }
    }
    const bool renorm = a.has_table ? a.hook_dim > 0 : a.normalize != 0;
    if (renorm) {
        double ss = 0.0;
        for (uint32_t k = 0; k < w; ++k) ss = fma(f[k], f[k], ss);
        const double inv = 1.0 / sqrt(ss);
        for (uint32_t k = 0; k < w; ++k) f[k] = inv * f[k];
    }
    if (a.has_table || a.ensure_pos) {
        for (uint32_t k = 0; k < w; ++k) f[k] = isq2 * f[k];
        f[w] = isq2 * 1.0;
    } else {
        f[w] = 0.0;
    }
    a.r[item] = exp(a.alpha * a.rel[item]);
}

// The common case of the above — table embeddings only, dim 64 or 128 — in one pass with the row in registers
// (16-B loads; the generic kernel walks its row three times through global memory, 63 us for 500 candidates against
// 35 us for the whole kernel matrix).  Same operations in the same order.
template <int D>
__global__ __launch_bounds__(64) void dpp_prepare_table_kernel(DppPrep a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = blockIdx.y;
    const size_t item = (size_t)q * a.n + (i < a.n ? i : a.n - 1);          // (lanes past the request's rows compute its last row again: every lane takes part in the LDS exchange below)
    const float4* x4 = reinterpret_cast<const float4*>(a.emb32 + item * D);
    double v[D];
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
        const float4 t = x4[k];
        v[4 * k] = (double)t.x;
        v[4 * k + 1] = (double)t.y;
        v[4 * k + 2] = (double)t.z;
        v[4 * k + 3] = (double)t.w;
    }
    if (a.normalize) {
        double ss = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) ss = fma(v[k], v[k], ss);
        const double inv = 1.0 / sqrt(ss);
#pragma unroll
        for (int k = 0; k < D; ++k) v[k] = inv * v[k];
    }
    const double isq2 = 0.70710678118654757;
    // F rows leave as whole 128-byte runs: the wave's 64 rows x 16 columns go through LDS, and a store instruction then covers
    // four rows x 16 columns (one lane per row wrote 8 bytes into 64 different lines per instruction: 129 x 64 line accesses per
    // wave — the kernel's 0.09 ms for 256 x 500 rows).  A wave is a workgroup: LDS operations execute in program order.
    __shared__ double tile[64][17];
    const uint32_t lane = threadIdx.x, kk = lane & 15, rr = lane >> 4;
    const size_t row0 = (size_t)q * a.n + (size_t)blockIdx.x * 64;           // the wave's first row (of this request)
    const uint32_t rows = a.n - blockIdx.x * 64 < 64u ? a.n - blockIdx.x * 64 : 64u;
    double* const Fw = a.F + row0 * (D + 1);
#pragma unroll
    for (int c = 0; c < D / 16; ++c) {
#pragma unroll
        for (int k = 0; k < 16; ++k) tile[lane][k] = isq2 * v[16 * c + k];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const uint32_t row = (uint32_t)it * 4 + rr;
            if (row < rows) Fw[(size_t)row * (D + 1) + 16 * c + kk] = tile[row][kk];
        }
    }
    if (i < a.n) {
        a.F[item * (D + 1) + D] = isq2 * 1.0;
        a.r[item] = exp(a.alpha * a.rel[item]);
    }
}

// S = F F^T (round 5: the scaling L = diag(r) S diag(r) is applied where the greedy kernels read) for one request: a 64 x 64 tile per workgroup, the two 64-row
// panels of F staged through LDS 16 columns at a time, every thread a 4 x 4 patch.  Each S_ij is still its own
// k-ascending fma chain (the staging only changes where the operands come from), so the bits are those of the
// one-thread-per-element version — which read every F row n times from L2: 66 GB for 256 requests x 500 candidates
// (cfg 5's batch), 30 ms; tiled it is ~1 ms.
// Which (request, tile pair) a workgroup of the kernel matrix takes.  The dispatcher places workgroup b on XCD b % 8, and every
// XCD has its own L2: with the tile pairs of a request dealt over the grid's x and the requests over z, the nt (nt + 1) / 2 tiles
// that share a request's F rows ran on all eight XCDs, and each fetched its two 64-row panels from the fabric again —
// FETCH_SIZE 1.75 GB per 256-request batch (profiles/r4_dpp_v2_pmc_summary.txt) for 132 MB of F, and 0.29 ms of the kernel's
// 0.35 with the arithmetic removed (round 5 ablation, DPP_ABL).  Here XCD x serves requests x, x + 8, …, all tile pairs of one
// request on consecutive workgroups of that XCD: a request's F (516 KB at 500 x 129) is fetched once and stays in the 4 MB L2
// while its tiles run.  (Placement is a speed assumption only: any other assignment of workgroups to XCDs computes the same.)
// Requests beyond the last full round of eight (and a batch of fewer than eight — one request of pg_dpp) are not left to one XCD
// each with the others idle: XCD x takes request x mod rem of that round and every ways-th tile pair of it, ways = the number of
// XCDs that share the request.
__device__ __forceinline__ bool dpp_tile_of_block(uint32_t nt, uint32_t R, uint32_t* q, uint32_t* p) {
    const uint32_t pairs = nt * (nt + 1) / 2, b = blockIdx.x;
    const uint32_t xcd = b & 7u, slot = b >> 3;
    const uint32_t full = R >> 3, rem = R & 7u, slots_full = full * pairs;
    if (slot < slots_full) {
        *q = xcd + 8u * (slot / pairs);
        *p = slot % pairs;
        return true;
    }
    if (rem == 0) return false;
    const uint32_t qq = xcd % rem, part = xcd / rem, ways = (8u - qq + rem - 1u) / rem;
    *q = 8u * full + qq;
    *p = part + ways * (slot - slots_full);
    return *p < pairs;
}
__host__ inline uint32_t dpp_km_blocks(uint32_t nt, uint32_t R) {
    const uint32_t pairs = nt * (nt + 1) / 2, full = R >> 3, rem = R & 7u;
    const uint32_t slots = full * pairs + (rem ? (pairs + 8u / rem - 1u) / (8u / rem) : 0u);     // (the request with the fewest XCDs: floor(8 / rem))
    return slots * 8u;
}

constexpr int kDppTile = 64, kDppKc = 16;
// One WAVE per 64 x 64 tile of S = F F^T, upper triangle only (blockIdx.x walks the tile pairs ti <= tj): lane (ty, tx)
// of an 8 x 8 grid owns the 8 x 8 patch rows {2ty, 2ty + 1} + 16a, columns {2tx, 2tx + 1} + 16b — 64 accumulators per
// lane, and an operand pair is ONE 16-B LDS read: 8 reads per 64 fma (the round-2 kernel's 4 x 4 patches needed 8
// reads per 16 fma and were LDS-bound at 28 % of the fp64 rate).  The panels are staged with 16 lanes per row (128
// contiguous bytes per row and instruction).  Every element is its own k-ascending fma chain, as the specification
// wants it; S is symmetric bit for bit (a product commutes), L is not — L_ij = (r_i S_ij) r_j and L_ji = (r_j S_ij) r_i
// are both formed from the one S_ij.
// Round 4 (profiles/r4_dpp_pmc_summary.txt, scripts/micro/fma64_rate.hip).  What the chip gives: v_fmac_f64 from 64
// independent accumulators runs at 4.7 cycles per wave-instruction and SIMD with two waves per SIMD (6.2 with one) at the
// 1.8 GHz it sustains under that load — 25 T fma/s, not the nominal 39: this kernel's 4.66 G fma cannot take less than
// 0.19 ms.  Where it stood: one wave per SIMD (496 registers), every chunk's panel loads awaited at the top of the chunk,
// VALU 38 % busy — 521 us.  Steps: next chunk's loads requested before the fma block (456 us); then TWO waves per SIMD —
// possible once the epilogue's passes were compile-time code (with `mirror` / `half` as loop variables the compiler kept
// the 64 accumulators in scratch and formed all 256 products in each of the four passes: 241 spilled dwords under a
// 256-register budget) and the panel loads moved back inside the chunk (their 64 registers are dead during the fma
// block; the other wave covers the wait): 365 us; panel loads without per-load arithmetic and a 17-wide last chunk
// instead of a ninth staging round for one column: 356 us = VALU 50 % busy at 2.1 GHz, 0.53 of what the chip gives.
// Tried and dropped: 8-column chunks double-buffered in LDS (700 us: a row's 64-B segments fetch every 128-B line twice);
// the operands of step k + 1 read before the fma of step k (no gain).
__global__ __launch_bounds__(64, 2) void dpp_kernel_matrix_kernel(const double* __restrict__ F,
                                                                  uint32_t n, uint32_t d1, uint32_t nt, uint32_t R, uint32_t ld, double* __restrict__ S) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    // the two panels [k][row], rows padded to a 16-B multiple (+ 1 column: the 17-wide tail); ONE array: the output staging
    // below runs over both
    __shared__ __attribute__((aligned(16))) double panels[2][kDppKc + 1][kDppTile + 2];
    auto& sa = panels[0];
    auto& sb = panels[1];
    // (request, tile pair) of this workgroup: see dpp_tile_of_block
    uint32_t q, p;
    if (!dpp_tile_of_block(nt, R, &q, &p)) return;
    // tile pair p → (ti, tj), ti <= tj: row ti holds nt - ti pairs
    uint32_t ti = 0;
    while (p >= nt - ti) {
        p -= nt - ti;
        ++ti;
    }
    const uint32_t tj = ti + p;
    const uint32_t i0 = ti * kDppTile, j0 = tj * kDppTile;
    const uint32_t lane = threadIdx.x, tx = lane & 7, ty = lane >> 3;
    const uint32_t kk = lane & 15, rr = lane >> 4;             // staging: column kk of rows rr + 4 it
    const double* Fq = F + (size_t)q * n * d1;
    double acc[8][8];                                          // [2a + u][2b + v]: row 2ty + u + 16a, column 2tx + v + 16b
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;
    double va[16], vb[16];
    // Panel loads with NO per-load arithmetic: a lane's byte offset (row rr, column k0 + kk) is one register advanced per
    // chunk, the sixteen row groups are wave-uniform bases (scalar registers).  Rows past n and columns past d1 are read
    // — F carries 64 rows of slack behind the last request (dpp_run_locked) — but only feed accumulators that are never
    // written / k-steps that are never taken.  (Clamped addresses cost 130 VALU instructions per chunk beside its 1 024 fma.)
    const char* const Fb = reinterpret_cast<const char*>(Fq);
    const uint32_t rowb = d1 * 8u;
    uint32_t voff = rr * rowb + kk * 8u;
    auto load_panels = [&]() {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const char* const ba = Fb + (size_t)(i0 + (uint32_t)it * 4) * rowb;      // (uniform)
            const char* const bb = Fb + (size_t)(j0 + (uint32_t)it * 4) * rowb;
            va[it] = *reinterpret_cast<const double*>(ba + voff);
            vb[it] = *reinterpret_cast<const double*>(bb + voff);
        }
        voff += kDppKc * 8u;
    };
    // a width of 16 m + 1 (the embedding's 128 columns + the constant one) ends with a chunk of 17 instead of a ninth
    // staging round for a single column
    const bool tail17 = d1 > (uint32_t)kDppKc && d1 % kDppKc == 1;
    for (uint32_t k0 = 0; k0 < d1; k0 += kDppKc) {
        uint32_t kc = d1 - k0 < (uint32_t)kDppKc ? d1 - k0 : (uint32_t)kDppKc;
        const bool last17 = tail17 && k0 + kDppKc + 1 == d1;
        load_panels();
        double xa = 0.0, xb = 0.0;
        if (last17) {                                          // column k0 + 16 of rows `lane` of both panels
            xa = *reinterpret_cast<const double*>(Fb + (size_t)(i0 + lane) * rowb + (size_t)(k0 + kDppKc) * 8u);
            xb = *reinterpret_cast<const double*>(Fb + (size_t)(j0 + lane) * rowb + (size_t)(k0 + kDppKc) * 8u);
        }
        __syncthreads();                                       // the previous step's readers are done
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            sa[kk][it * 4 + rr] = va[it];
            sb[kk][it * 4 + rr] = vb[it];
        }
        if (last17) {
            sa[kDppKc][lane] = xa;
            sb[kDppKc][lane] = xb;
            kc = kDppKc + 1;
        }
        __syncthreads();
        for (uint32_t k = 0; k < kc; ++k) {
            f64x2 av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = *reinterpret_cast<const f64x2*>(&sa[k][2 * ty + 16 * a]);
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = *reinterpret_cast<const f64x2*>(&sb[k][2 * tx + 16 * b]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[2 * a][2 * b] = fma(av[a].x, bv[b].x, acc[2 * a][2 * b]);
                    acc[2 * a][2 * b + 1] = fma(av[a].x, bv[b].y, acc[2 * a][2 * b + 1]);
                    acc[2 * a + 1][2 * b] = fma(av[a].y, bv[b].x, acc[2 * a + 1][2 * b]);
                    acc[2 * a + 1][2 * b + 1] = fma(av[a].y, bv[b].y, acc[2 * a + 1][2 * b + 1]);
                }
        }
        if (last17) break;
    }
    // The S tile and (off the diagonal) its mirror — its transpose; round 5: the r-scaling moved to the greedy kernels — written as whole 512-B rows: the patches go through LDS — half a tile
    // (32 rows) at a time, in the panels' space — so that a store instruction covers one contiguous row of 64 doubles
    // (patch-wise stores are 16-B runs scattered over eight rows: 512 MB of them per 256-request batch)
    double* const stage = &panels[0][0][0];                     // 32 x 65 doubles fit the two panels (2 x 17 x 66)
    static_assert(32 * 65 <= 2 * (kDppKc + 1) * (kDppTile + 2), "the output staging aliases the panels");
    // (mirror and half as compile-time values: with runtime ones the compiler keeps the accumulators in scratch and forms
    // every product in every pass)
    auto pass = [&](auto mirror_c, auto half_c) {
        constexpr int mirror = decltype(mirror_c)::value, half = decltype(half_c)::value;
        __syncthreads();                                    // the panels' / the previous half's readers are done
        // this lane's elements whose OUTPUT row falls into rows [32 half, 32 half + 32) of the (mirrored) tile
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int row_t = 2 * (int)ty + (a & 1) + 16 * (a >> 1);        // row of the tile
                const int col_t = 2 * (int)tx + (b & 1) + 16 * (b >> 1);
                const int orow = mirror ? col_t : row_t, ocol = mirror ? row_t : col_t;
                if (((mirror ? b : a) >> 2) != half) continue;                  // (orow >> 5: 2 t + (x & 1) < 16)
                stage[(orow & 31) * 65 + ocol] = acc[a][b];
            }
        __syncthreads();
        const uint32_t r0 = (mirror ? j0 : i0) + 32 * half, c0 = mirror ? i0 : j0;
        const uint32_t rows = r0 < n ? (n - r0 < 32u ? n - r0 : 32u) : 0u;      // (uniform; the column test once per pass)
        double* const rowp = S + ((size_t)q * n + r0) * ld + c0;                 // (uniform: stores with a scalar base)
        if (c0 + lane < n)
            for (uint32_t rr2 = 0; rr2 < rows; ++rr2)
                __builtin_nontemporal_store(stage[rr2 * 65 + lane], rowp + (size_t)rr2 * ld + lane);   // (streams out: see the matrix-pipe kernel)
    };
    pass(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
    pass(std::integral_constant<int, 0>(), std::integral_constant<int, 1>());
    if (ti != tj) {
        pass(std::integral_constant<int, 1>(), std::integral_constant<int, 0>());
        pass(std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
    }
}

// Round 5: the same tile on the fp64 matrix pipe.  v_mfma_f64_16x16x4_f64 accumulates D = C + a_0 b_0 + … + a_3 b_3 as the
// k-ascending fma chain the specification is (scripts/micro/mfma_f64.hip, profiles/r5_mfma_f64_microbench.txt: 256 of 256
// outputs over K = 128 with exponents spread over 2^±20 bit-equal to the chain; 125 / 46 / 65 of 256 to the other candidate
// orders), so S_ij keeps its bits (tests/test_gpu_parity.py::test_dpp_kernel_matrix_bits_on_both_pipes: this kernel, the vector
// kernel and the oracle agree bit for bit) — and an instruction retires 1 024 fma for TWO 8-byte operand reads per lane where the
// vector form needs 8 reads of 16 B per 64.  The instruction takes 64 cycles (16 fma per clock and SIMD = 38 T fma/s measured:
// the chip's fp64 matrix peak IS its vector peak — the instructions execute on the SIMD's fp64 vector units), 33 k-steps x 16
// blocks per tile = 0.136 ms of pipe time per 256 x 500 x 129 batch.  A wave owns the 64 x 64 tile as 4 x 4 blocks of 16 x 16:
// block (a, b), register g, lane l = row 16a + (l >> 4) + 4g, column 16b + (l & 15).  The panels are staged as before, the next
// chunk's requested before this chunk's instructions; a width that is not a multiple of four ends with zero operands written
// into the panel (fma(0, 0, acc) = acc: the chain starts at +0 and cannot reach -0).  Epilogue: the tile of S and its mirror (its
// transpose: S is symmetric bit for bit) as whole rows through LDS; L_ij = (r_i S_ij) r_j is formed by the greedy kernels for the
// diagonal and the rows they pick — a fifth of the matrix — which took 1 024 multiplications per tile out of this epilogue.
// Measured (profiles/r5_dpp_mfma_summary.txt): 0.343 ms as first built — exactly the vector kernel's time: neither pipe was the
// bound.  FETCH_SIZE 1.75 GB for 132 MB of F: dpp_tile_of_block (one request's tiles on one XCD) → 135 MB, 0.304 ms; L's 512 MB as
// non-temporal stores: 0.277 ms (matrix pipe 49 % busy); S instead of L, the stores' tests hoisted: 0.262 ms; panels staged with 16-byte
// loads (eight lanes per row, half the load instructions): −5 % (0.258 against 0.274 on one box).  What is left is the pairing of two waves per SIMD on one fp64 unit: with
// the matrix instructions removed the kernel takes 0.144 ms, with them 0.277 = the sum, not the maximum — a wave's staging and
// epilogue (≈ 1 700 vector instructions per tile) advance at about one instruction per partner matrix instruction (64 cycles),
// s_setprio does not change that, and one wave per SIMD (0.37 ms) leaves every load latency exposed.  Not done: a single wave
// per SIMD that issues its LDS / global traffic between its own matrix instructions (hand-placed, as csrc/recall.hip does).
__global__ __launch_bounds__(64, 2) void dpp_kernel_matrix_mfma_kernel(const double* __restrict__ F,
                                                                       uint32_t n, uint32_t d1, uint32_t nt, uint32_t R, uint32_t ld, double* __restrict__ S
#ifdef DPP_PROFILE
                                                                       , unsigned long long* prof
#endif
                                                                       ) {
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) double panels[2][kDppKc + 1][kDppTile + 2];
    auto& sa = panels[0];
    auto& sb = panels[1];
    uint32_t q, p;
    if (!dpp_tile_of_block(nt, R, &q, &p)) return;
#ifdef DPP_PROFILE
    // phases: 0 prologue (first panels requested), 1 waiting for a chunk's panels + their LDS writes, 2 requesting the next panels,
    // 3 the chunk's matrix instructions (operand reads included), 4 epilogue
    uint64_t ph[5] = {0, 0, 0, 0, 0}, tp = __builtin_readcyclecounter();
#define DPP_MARK(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const uint64_t tn = __builtin_readcyclecounter(); ph[i] += tn - tp; tp = tn; }
#else
#define DPP_MARK(i)
#endif
    uint32_t ti = 0;
    while (p >= nt - ti) {
        p -= nt - ti;
        ++ti;
    }
    const uint32_t tj = ti + p;
    const uint32_t i0 = ti * kDppTile, j0 = tj * kDppTile;
    const uint32_t lane = threadIdx.x;
    const uint32_t kk = lane & 15, rr = lane >> 4;             // staging: column kk of rows rr + 4 it; operands: row kk, k-slot rr
    const double* Fq = F + (size_t)q * n * d1;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
#define DPP_ACCS "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), \
                 "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3])
    asm volatile("s_nop 3" : DPP_ACCS);                         // the zeros are VALU writes
    // One wave per workgroup: LDS accesses of a wave execute in program order, so the panels need no barrier — and must not have
    // __syncthreads(), whose s_waitcnt vmcnt(0) would make every chunk wait for the NEXT chunk's panel loads, which are requested
    // before the chunk's instructions so that they land under them (and, in the epilogue, for the previous pass's stores).
    // staging: 16 bytes per lane — columns 2 kq, 2 kq + 1 of rows r8 + 8 it (eight lanes per row: its chunk's 128 bytes)
    typedef double f64x2 __attribute__((ext_vector_type(2), aligned(8)));
    f64x2 va[8], vb[8];
    const char* const Fb = reinterpret_cast<const char*>(Fq);
    const uint32_t rowb = d1 * 8u;
    const uint32_t kq = lane & 7, r8 = lane >> 3;
    uint32_t voff = r8 * rowb + kq * 16u;
    auto load_panels = [&]() {                                  // (unclamped: see dpp_kernel_matrix_kernel)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            va[it] = *reinterpret_cast<const f64x2*>(Fb + (size_t)(i0 + (uint32_t)it * 8) * rowb + voff);
            vb[it] = *reinterpret_cast<const f64x2*>(Fb + (size_t)(j0 + (uint32_t)it * 8) * rowb + voff);
        }
        voff += kDppKc * 8u;
    };
    // (inline asm with the accumulator as "+v": the builtin let the allocator place a k-step's results in fresh registers — 271
    // spilled dwords under the two-waves budget.  Wait states by hand: operands fresh from LDS → s_nop 3; the accumulators are
    // read by nothing but their own next instruction until the s_nops behind the loop.)  No selects: fp64 matrix instructions
    // execute on the SIMD's fp64 vector units, and beside a partner wave that issues them back to back every OTHER vector
    // instruction of a wave gets one slot per 64 cycles, whatever its priority (round 5 profile build: 150 cycles per panel load,
    // 90 K cycles for the epilogue's ~1 000 instructions) — k-slots past the width read zeros that were written into the panels.
    auto step = [&](uint32_t kr) {                              // one instruction per block: k-slot kr of this lane
        double av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            av[a] = sa[kr][16 * a + kk];
            bv[a] = sb[kr][16 * a + kk];
        }
        asm volatile("s_nop 3" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[a][b]) : "v"(av[a]), "v"(bv[b]));
    };
    load_panels();
    DPP_MARK(0)
    for (uint32_t k0 = 0; k0 < d1; k0 += kDppKc) {              // (the 129th column is a ninth chunk of one: its staging hides like the others')
        const uint32_t kc = d1 - k0 < (uint32_t)kDppKc ? d1 - k0 : (uint32_t)kDppKc;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            sa[2 * kq][it * 8 + r8] = va[it].x;
            sa[2 * kq + 1][it * 8 + r8] = va[it].y;
            sb[2 * kq][it * 8 + r8] = vb[it].x;
            sb[2 * kq + 1][it * 8 + r8] = vb[it].y;
        }
        for (uint32_t z = kc; z < ((kc + 3u) & ~3u); ++z) {     // a width that is not a multiple of four: zeros behind it
            sa[z][lane] = 0.0;
            sb[z][lane] = 0.0;
        }
        DPP_MARK(1)
        if (k0 + kDppKc < d1) load_panels();                    // the next chunk's, under this chunk's instructions
        DPP_MARK(2)
#if DPP_ABL == 2
        if (d1 == 12345u)
#endif
#pragma unroll 1
        for (uint32_t s4 = 0; 4 * s4 < kc; ++s4) step(4 * s4 + rr);   // (one code path: the accumulators stay where they are)
        asm volatile("" ::: "memory");
        DPP_MARK(3)
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : DPP_ACCS);   // a 16-pass instruction's results, before anything else reads them
#undef DPP_ACCS
    double* const stage = &panels[0][0][0];                     // 32 x 65 doubles fit the two panels (2 x 17 x 66)
    static_assert(32 * 65 <= 2 * (kDppKc + 1) * (kDppTile + 2), "the output staging aliases the panels");
    auto pass = [&](auto mirror_c, auto half_c) {
        constexpr int mirror = decltype(mirror_c)::value, half = decltype(half_c)::value;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (((mirror ? b : a) >> 1) != half) continue;                  // output rows [32 half, 32 half + 32)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row_t = 16 * a + (int)rr + 4 * g, col_t = 16 * b + (int)kk;
                    const int orow = mirror ? col_t : row_t, ocol = mirror ? row_t : col_t;
                    stage[(orow & 31) * 65 + ocol] = acc[a][b][g];
                }
            }
        const uint32_t r0 = (mirror ? j0 : i0) + 32 * half, c0 = mirror ? i0 : j0;
        const uint32_t rows = r0 < n ? (n - r0 < 32u ? n - r0 : 32u) : 0u;      // (uniform; the column test once per pass)
        double* const rowp = S + ((size_t)q * n + r0) * ld + c0;                 // (uniform: stores with a scalar base)
#if DPP_ABL == 1
        if (c0 + lane < n && stage[lane] == 1.2345e300)
#else
        if (c0 + lane < n)
#endif
            for (uint32_t rr2 = 0; rr2 < rows; ++rr2)
                // (non-temporal: 2 MB per request stream out once; as ordinary stores they pass through the L2 the tiles' F rows live in)
                __builtin_nontemporal_store(stage[rr2 * 65 + lane], rowp + (size_t)rr2 * ld + lane);
    };
    pass(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
    pass(std::integral_constant<int, 0>(), std::integral_constant<int, 1>());
    if (ti != tj) {
        pass(std::integral_constant<int, 1>(), std::integral_constant<int, 0>());
        pass(std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
    }
#ifdef DPP_PROFILE
    DPP_MARK(4)
    if (lane == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(&prof[i], (unsigned long long)ph[i]);
#endif
#undef DPP_MARK
}

// L_ij = (r_i S_ij) r_j for the whole matrix (pg_dpp_kernel_matrix_dev; the greedy kernels form the elements they read themselves)
__global__ void dpp_scale_kernel(const double* __restrict__ S, const double* __restrict__ r, uint32_t n, uint32_t ld,
                                 double* __restrict__ L) {
    const uint32_t j = blockIdx.x * 64u + threadIdx.x, i = blockIdx.y, q = blockIdx.z;
    if (j >= n) return;
    const double* rq = r + (size_t)q * n;
    L[((size_t)q * n + i) * n + j] = __dmul_rn(__dmul_rn(rq[i], S[((size_t)q * n + i) * ld + j]), rq[j]);
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
//   One workgroup per request (blockIdx.x): L, d2, c, out are that request's slices.
__global__ __launch_bounds__(1024) void dpp_greedy_kernel(const double* __restrict__ S_all, const double* __restrict__ r_all, uint32_t N, uint32_t ld,
                                                          uint32_t topn_total, uint32_t window,
                                                          double* __restrict__ d2_all, double* __restrict__ c_all,
                                                          uint32_t* __restrict__ out_all, uint32_t* __restrict__ out_count) {
    const uint32_t req = blockIdx.x;
    const uint32_t wrows = window < N ? window : N;
    const double* __restrict__ S = S_all + (size_t)req * N * ld;          // rows of ld doubles (dpp_run_locked)
    const double* __restrict__ r = r_all + (size_t)req * N;
    // L_ij = (r_i S_ij) r_j (dpp_sort.go:463-472), formed where an element is read: the kernel matrix stores S
    auto L_at = [&](uint32_t i, uint32_t j) { return __dmul_rn(__dmul_rn(r[i], S[(size_t)i * ld + j]), r[j]); };
    double* __restrict__ d2 = d2_all + (size_t)req * N;
    double* __restrict__ c = c_all + (size_t)req * wrows * N;
    uint32_t* __restrict__ out = out_all + (size_t)req * topn_total;
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
            d2[i] = ex ? nan : L_at(i, i);
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
                double lj = L_at(j, n);
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
    if (tid == 0) out_count[req] = done;
}

// The same greedy inference for candidate sets of up to EPL x 64 items and windows of up to WMAX picks, ONE WAVE per
// request: lane l owns items l, l + 64, ... — their d2 and their columns of c live in registers (statically indexed:
// the picks of a window are unrolled), the picked item's column c[.][j] is read back from an LDS copy (broadcast).
// No workgroup barrier, no tree reduction through LDS: a pick is one row of L from L2 (issued first, the c dot
// products run under it), EPL x k multiply-adds and a butterfly argmax.  The workgroup version above spends ~20
// barriers per pick: 100 picks of 500 candidates took 0.35 ms for one request (1.5 ms for each of 256 concurrent
// ones); the arithmetic, its order and the tie rules are the same, so the picks are identical.
template <int K0, int K1, class F>
__device__ __forceinline__ void dpp_static_for(F&& f) {
    if constexpr (K0 < K1) {
        f(std::integral_constant<int, K0>{});
        dpp_static_for<K0 + 1, K1>(f);
    }
}

template <int EPL, int WMAX>
__global__ __launch_bounds__(64) void dpp_greedy_wave_kernel(const double* __restrict__ S_all, const double* __restrict__ r_all, uint32_t N, uint32_t ld,
                                                             uint32_t topn_total, uint32_t window,
                                                             uint32_t* __restrict__ out_all, uint32_t* __restrict__ out_count) {
    extern __shared__ double c_lds[];                   // [min(window, WMAX)][EPL * 64]
    constexpr uint32_t NS = EPL * 64;
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t req = blockIdx.x, lane = threadIdx.x;
    const double* __restrict__ S = S_all + (size_t)req * N * ld;          // rows of ld doubles (dpp_run_locked)
    const double* __restrict__ r = r_all + (size_t)req * N;
    uint32_t* __restrict__ out = out_all + (size_t)req * topn_total;
    const double epsilon = 1e-10;
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    // L_ij = (r_i S_ij) r_j (dpp_sort.go:463-472), formed where an element is read — the diagonal and the picked rows, a fifth of
    // the matrix: the kernel matrix stores S, and its epilogue lost the 4 multiplications per element pair
    double rn[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const uint32_t n = (uint32_t)s * 64u + lane;
        rn[s] = n < N ? r[n] : 0.0;
    }
    double c[EPL][WMAX];
    double d2[EPL];
    bool sel[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) sel[s] = false;
    // floats.MaxIdx over d2: first maximum, NaN skipped, nothing left → index 0; returns d2[j] as well.  The reduction is in every
    // pick's dependent chain (round 5 profile build: 1 750 of a pick's 5 500 cycles as a butterfly over (value, index) pairs), so
    // it is split: the wave's maximum VALUE (v_max_f64 passes over NaN) through six lane exchanges, then the first index that
    // holds it from one ballot per element slot — scalar work.  Both results are made wave-uniform registers: the window's
    // `stop` / `broke` tests then compile to scalar branches instead of exec-mask bookkeeping around every unrolled pick.
    auto argmax = [&](uint32_t& j, double& dj) {
        double m = d2[0];
#pragma unroll
        for (int s = 1; s < EPL; ++s) m = fmax(m, d2[s]);
        dpp_static_for<0, 6>([&](auto sc) {
            constexpr int off = 1 << decltype(sc)::value;
            const uint64_t bb = (uint64_t)__double_as_longlong(m);
            const uint32_t lo = lane_xor<off>((uint32_t)bb), hi = lane_xor<off>((uint32_t)(bb >> 32));
            m = fmax(m, __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)));
        });
        auto uniform = [](double x) {
            const uint64_t b = (uint64_t)__double_as_longlong(x);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
            return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        };
        m = uniform(m);
        uint32_t found = kNone;
#pragma unroll
        for (int s = EPL - 1; s >= 0; --s) {                    // (descending: the lowest slot that holds the maximum is assigned last)
            const uint64_t b = __builtin_amdgcn_ballot_w64(d2[s] == m);
            if (b) found = (uint32_t)s * 64u + (uint32_t)__builtin_ctzll(b);
        }
        if (found == kNone) {                                   // nothing but NaN
            j = 0u;
            dj = uniform(d2[0]);                                // (lane 0's: element 0)
        } else {
            j = found;
            dj = m;
        }
    };
    uint32_t done = 0;
#ifdef DPP_PROFILE
    uint64_t prof_wait = 0, prof_picks = 0, prof_arg = 0, prof_upd = 0;
    const uint64_t prof_t0 = __builtin_readcyclecounter();
    uint64_t prof_mark = prof_t0;
#endif
    uint32_t n_calls, rem;
    if (topn_total <= window) { n_calls = 1; rem = 0; }
    else { n_calls = topn_total / window; rem = topn_total % window; }
    for (uint32_t call = 0; call < n_calls + (rem ? 1u : 0u); ++call) {
        uint32_t topn = (topn_total <= window) ? topn_total : (call < n_calls ? window : rem);
        if (topn > N) topn = N;
        if (topn == 0) continue;
        {
            double sv[EPL];                                     // (every load first, unconditionally: a select per element made the
#pragma unroll                                                  //  compiler branch around each load and wait for them one by one)
            for (int s = 0; s < EPL; ++s) {
                const uint32_t n = (uint32_t)s * 64u + lane, nc = n < N ? n : 0u;
                sv[s] = S[(size_t)nc * ld + nc];
            }
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                const uint32_t n = (uint32_t)s * 64u + lane;
                const double l = __dmul_rn(__dmul_rn(rn[s], sv[s]), rn[s]);
                d2[s] = (n < N && !sel[s]) ? l : nan;           // already selected (and the padding): NaN
            }
        }
        uint32_t j;
        double dj;
        argmax(j, dj);
        if (lane == 0) out[done] = j;
#pragma unroll
        for (int s = 0; s < EPL; ++s) sel[s] = sel[s] || ((uint32_t)s * 64u + lane == j);
        uint32_t ny = 1;
        bool broke = false;
        bool stop = false;
        dpp_static_for<0, WMAX>([&](auto kc) {           // k = ny - 1: the picks of one window, unrolled
            constexpr int k = decltype(kc)::value;
            if (stop) return;
            if (ny >= topn) { stop = true; return; }
            if (dj < epsilon) { broke = true; stop = true; return; }
            double lv[EPL];
#pragma unroll
            for (int s = 0; s < EPL; ++s) {                      // (the row's loads first, all of them: see above)
                const uint32_t n = (uint32_t)s * 64u + lane;
                lv[s] = S[(size_t)j * ld + (n < N ? n : 0u)];
            }
            double cj[WMAX];
#pragma unroll
            for (int i = 0; i < k; ++i) cj[i] = c_lds[(uint32_t)i * NS + j];
            __builtin_amdgcn_sched_barrier(0);                   // … and the earlier picks' column entries: requested before the
                                                                 // square root and the division below, whose ~40 dependent fp64 operations they land under
            const double inv = 1.0 / sqrt(dj);
            double rsel = rn[0];                                 // r[j] from the lane that holds it (j is wave-uniform): no load in the pick's chain
#pragma unroll
            for (int s = 1; s < EPL; ++s) rsel = (j >> 6) == (uint32_t)s ? rn[s] : rsel;
            const double rj = __shfl(rsel, (int)(j & 63u), 64);
#ifdef DPP_PROFILE
            {                                                    // cycles from the row's requests to its arrival
                const uint64_t t0 = __builtin_readcyclecounter();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                prof_wait += __builtin_readcyclecounter() - t0;
                ++prof_picks;
            }
#endif
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                const uint32_t n = (uint32_t)s * 64u + lane;
                const double l = __dmul_rn(__dmul_rn(rj, lv[s]), rn[s]);
                lv[s] = n < N ? l : 0.0;
            }
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                double lj = lv[s];
                if (k > 0) {
                    double ss = 0.0;
#pragma unroll
                    for (int i = 0; i < k; ++i) ss = __dadd_rn(ss, __dmul_rn(cj[i], c[s][i]));
                    lj = __dsub_rn(lj, ss);
                }
                const double e = __dmul_rn(inv, lj);
                c[s][k] = e;
                d2[s] = __dsub_rn(d2[s], __dmul_rn(e, e));
                c_lds[(uint32_t)k * NS + (uint32_t)s * 64u + lane] = e;
                if ((uint32_t)s * 64u + lane == j) d2[s] = nan;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: orders the LDS column copy for the later picks
#ifdef DPP_PROFILE
            const uint64_t ta0 = __builtin_readcyclecounter();
            prof_upd += ta0 - prof_mark;
#endif
            argmax(j, dj);
#ifdef DPP_PROFILE
            asm volatile("" : "+v"(dj));
            prof_mark = __builtin_readcyclecounter();
            prof_arg += prof_mark - ta0;
#endif
            if (lane == 0) out[done + ny] = j;
#pragma unroll
            for (int s = 0; s < EPL; ++s) sel[s] = sel[s] || ((uint32_t)s * 64u + lane == j);
            ++ny;
        });
        if (broke && ny < topn) {
            // the rest of the window by index, skipping what is already in the result (dpp_sort.go:541-548)
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                const uint32_t n = (uint32_t)s * 64u + lane;
                const bool cand = n < N && !sel[s];
                const uint64_t m = __builtin_amdgcn_ballot_w64(cand);
                const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t room = topn - ny, cnt = (uint32_t)__popcll(m);
                if (cand && before < room) {
                    out[done + ny + before] = n;
                    sel[s] = true;
                }
                ny += cnt < room ? cnt : room;
            }
        }
        done += ny;
    }
    if (lane == 0) out_count[req] = done;
#ifdef DPP_PROFILE
    if (lane == 0 && req == 3)
        printf("dpp greedy (request 3): %llu picks, %llu ticks in all, %llu waiting for the picked rows, %llu in the argmax, %llu from one argmax to the next (row wait included)\n",
               (unsigned long long)prof_picks, (unsigned long long)(__builtin_readcyclecounter() - prof_t0), (unsigned long long)prof_wait,
               (unsigned long long)prof_arg, (unsigned long long)prof_upd);
#endif
}

// DPP for R independent requests of n candidates each, device-resident: d_emb32 [R][n][d] fp32 (NULL on the
// hook-only path), d_hook [R][n][hook_dim] fp64 (or NULL), d_rel [R][n] relevance scores as KernelMatrix uses
// them (already normalised when dpp_norm_relevance_score is on); d_out [R][topn] candidate indices, d_out_count [R].
// Caller holds ctx->mu; nothing synchronises.
int dpp_run_locked(pg_ctx* ctx, const float* d_emb32, const double* d_hook, const double* d_rel, uint32_t R, uint32_t n,
                   uint32_t d, uint32_t hook_dim, double alpha, uint32_t topn, uint32_t window, int normalize,
                   int ensure_pos, int has_table, uint32_t* d_out, uint32_t* d_out_count, double* d_L_out) {
    if (R == 0 || n == 0 || (topn == 0 && !d_L_out)) return PG_OK;
    if (window == 0) window = 10;                        // NewDPPSort default (dpp_sort.go:89-91)
    const uint32_t d1 = hook_dim + (has_table ? d : 0u) + 1;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // L's rows are padded to whole 128-byte lines: a tile's 512-byte row segments then cover four lines each — with rows of n = 500
    // doubles every segment began and ended inside a line that a workgroup of another XCD completes, and the 512 MB of a
    // 256-request batch left the chip as partial-line writes at 1.46 TB/s (0.35 ms whatever the arithmetic pipe did)
    const uint32_t ld = (n + 15u) & ~15u;
    // (F: 64 rows of slack behind the last request — the kernel matrix's panel loads are not clamped)
    const size_t bF = al(((size_t)R * n + kDppTile) * d1 * 8), bR = al((size_t)R * n * 8), bL = al((size_t)R * n * ld * 8);
    const size_t bD2 = al((size_t)R * n * 8), bC = al((size_t)R * std::min(window, n) * n * 8);
    void* buf;
    int rc;
    if ((rc = scratch_reserve(ctx, 7, bF + bR + bL + bD2 + bC, &buf))) return rc;
    char* p = (char*)buf;
    double* F = (double*)p; p += bF;
    double* Rr = (double*)p; p += bR;
    double* L = (double*)p; p += bL;
    double* D2 = (double*)p; p += bD2;
    double* Cm = (double*)p;
    DppPrep a;
    a.emb32 = d_emb32; a.hook = d_hook; a.rel = d_rel;
    a.n = n; a.d = d; a.hook_dim = hook_dim; a.alpha = alpha;
    a.normalize = normalize; a.ensure_pos = ensure_pos; a.has_table = has_table;
    a.F = F; a.r = Rr;
    if (has_table && hook_dim == 0 && d == 128) dpp_prepare_table_kernel<128><<<dim3((n + 63) / 64, R), 64, 0, ctx->stream>>>(a);
    else if (has_table && hook_dim == 0 && d == 64) dpp_prepare_table_kernel<64><<<dim3((n + 63) / 64, R), 64, 0, ctx->stream>>>(a);
    else dpp_prepare_kernel<<<dim3((n + 63) / 64, R), 64, 0, ctx->stream>>>(a);
    const uint32_t nt = (n + kDppTile - 1) / kDppTile;
    const uint32_t km_blocks = dpp_km_blocks(nt, R);                            // (dpp_tile_of_block)
    if (ctx->knobs.dpp_valu) dpp_kernel_matrix_kernel<<<km_blocks, 64, 0, ctx->stream>>>(F, n, d1, nt, R, ld, L);
#ifdef DPP_PROFILE
    else {
        static unsigned long long* prof = nullptr;
        if (!prof) hipMalloc(&prof, 5 * 8);
        hipMemsetAsync(prof, 0, 5 * 8, ctx->stream);
        dpp_kernel_matrix_mfma_kernel<<<km_blocks, 64, 0, ctx->stream>>>(F, n, d1, nt, R, ld, L, prof);
        unsigned long long h[5];
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(h, prof, sizeof h, hipMemcpyDeviceToHost);
        static int calls = 0;
        if (++calls == 5) {
            const double tiles = (double)(nt * (nt + 1) / 2) * R;
            fprintf(stderr, "dpp kernel matrix, s_memtime ticks per tile: prologue %.0f, panel wait + LDS writes %.0f, panel requests %.0f, matrix instructions %.0f, epilogue %.0f\n",
                    h[0] / tiles, h[1] / tiles, h[2] / tiles, h[3] / tiles, h[4] / tiles);
        }
    }
#else
    else dpp_kernel_matrix_mfma_kernel<<<km_blocks, 64, 0, ctx->stream>>>(F, n, d1, nt, R, ld, L);
#endif
    if (d_L_out) {                                              // KernelMatrix alone (pg_dpp_kernel_matrix_dev)
        PG_HIP(hipGetLastError());
        dpp_scale_kernel<<<dim3((n + 63) / 64, n, R), 64, 0, ctx->stream>>>(L, Rr, n, ld, d_L_out);
        return PG_OK;
    }
    const uint32_t wrows = window < n ? window : n;
    if (n <= 512 && window <= 16) {
        const size_t lds = (size_t)wrows * 512 * 8;
        if ((rc = ensure_dyn_lds(ctx, (const void*)dpp_greedy_wave_kernel<8, 16>, lds))) return rc;
        dpp_greedy_wave_kernel<8, 16><<<R, 64, lds, ctx->stream>>>(L, Rr, n, ld, topn, window, d_out, d_out_count);
    } else if (n <= 1024 && window <= 10) {
        const size_t lds = (size_t)wrows * 1024 * 8;
        if ((rc = ensure_dyn_lds(ctx, (const void*)dpp_greedy_wave_kernel<16, 10>, lds))) return rc;
        dpp_greedy_wave_kernel<16, 10><<<R, 64, lds, ctx->stream>>>(L, Rr, n, ld, topn, window, d_out, d_out_count);
    } else {
        dpp_greedy_kernel<<<R, 1024, 0, ctx->stream>>>(L, Rr, n, ld, topn, window, D2, Cm, d_out, d_out_count);
    }
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// dpp_norm_relevance_score (dpp_sort.go:382-405): O(n) scalar work in the reference's operation order
// (stat.PopMeanVariance two-pass with compensation, stat.StdScore; min-max takes max = first, min = last item: the
// candidates arrive sorted by score).  One implementation for the host (a caller's thread) and the device (one thread
// per request inside a batch): IEEE fp64 add / mul / div / sqrt, no contraction, so both give the same bits.
__host__ __device__ inline bool dpp_norm_relevance_one(const double* rel, uint32_t n, int mode, double* out) {
#ifdef __HIP_DEVICE_COMPILE__
#define PG_DMUL(a, b) __dmul_rn((a), (b))
#define PG_DADD(a, b) __dadd_rn((a), (b))
#else
#define PG_DMUL(a, b) ([](double x_, double y_) { volatile double r_ = x_ * y_; return (double)r_; }((a), (b)))
#define PG_DADD(a, b) ([](double x_, double y_) { volatile double r_ = x_ + y_; return (double)r_; }((a), (b)))
#endif
    if (mode == 1) {
        double sum = 0.0;
        for (uint32_t i = 0; i < n; ++i) sum = PG_DADD(sum, rel[i]);
        const double mean = sum / (double)n;
        double ss = 0.0, comp = 0.0;
        for (uint32_t i = 0; i < n; ++i) {
            const double d = PG_DADD(rel[i], -mean);
            ss = PG_DADD(ss, PG_DMUL(d, d));
            comp = PG_DADD(comp, d);
        }
        const double variance = PG_DADD(ss, -(PG_DMUL(comp, comp) / (double)n)) / (double)n;
        if (mean == 0.0 || variance == 0.0) return false;
        const double sd = sqrt(variance);
        for (uint32_t i = 0; i < n; ++i) out[i] = PG_DADD(rel[i], -mean) / sd;
        return true;
    }
    if (mode == 2) {
        const double mx = rel[0], mn = rel[n - 1], span = PG_DADD(mx, -mn);
        if (span == 0.0) return false;
        const double eps = 1e-6;
        for (uint32_t i = 0; i < n; ++i) out[i] = PG_DADD(PG_DMUL(PG_DADD(rel[i], -mn) / span, 1 - eps), eps);
        return true;
    }
    if (out != rel)
        for (uint32_t i = 0; i < n; ++i) out[i] = rel[i];
    return true;
#undef PG_DMUL
#undef PG_DADD
}

bool dpp_norm_relevance_host(const double* rel, uint32_t n, int mode, double* out) {
    return dpp_norm_relevance_one(rel, n, mode, out);
}

__global__ void dpp_norm_relevance_kernel(double* __restrict__ rel, uint32_t nq, uint32_t n, int mode,
                                          uint32_t* __restrict__ bail) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    bail[q] = dpp_norm_relevance_one(rel + (size_t)q * n, n, mode, rel + (size_t)q * n) ? 0u : 1u;
}

__global__ void dpp_bail_fix_kernel(const uint32_t* __restrict__ bail, uint32_t nq, uint32_t n, uint32_t top_n,
                                    uint32_t* __restrict__ pick, uint32_t* __restrict__ pick_cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * top_n) return;
    const uint32_t q = i / top_n, p = i - q * top_n;
    if (!bail[q]) return;
    pick[i] = p;
    if (p == 0) pick_cnt[q] = top_n < n ? top_n : n;
}

int dpp_norm_relevance_launch(hipStream_t st, double* d_rel, uint32_t nq, uint32_t n, int mode, uint32_t* d_bail) {
    if (nq == 0) return PG_OK;
    if (mode == 0) {
        PG_HIP(hipMemsetAsync(d_bail, 0, (size_t)nq * 4, st));
        return PG_OK;
    }
    dpp_norm_relevance_kernel<<<(nq + 63) / 64, 64, 0, st>>>(d_rel, nq, n, mode, d_bail);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

int dpp_bail_fix_launch(hipStream_t st, const uint32_t* d_bail, uint32_t nq, uint32_t n, uint32_t top_n, uint32_t* d_pick,
                        uint32_t* d_pick_cnt) {
    if (nq == 0 || top_n == 0) return PG_OK;
    dpp_bail_fix_kernel<<<(nq * top_n + 255) / 256, 256, 0, st>>>(d_bail, nq, n, top_n, d_pick, d_pick_cnt);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // namespace pg

namespace pg {
int table_gather_locked(pg_ctx* ctx, const pg_table* t, const uint32_t* d_rows, uint32_t n, float* d_out);   // table.hip
}

extern "C" {

int pg_dpp_ex(pg_ctx* ctx, const pg_table* t, const uint32_t* cand_rows, const double* rel, uint32_t n,
              const pg_dpp_options* o, const double* hook_emb, uint32_t* out_idx, uint32_t* out_count,
              double* out_relevance) {
    PG_REQUIRE(ctx && o && out_count, "pg_dpp_ex: NULL argument");
    *out_count = 0;
    if (n == 0 || o->topn == 0) return PG_OK;
    PG_REQUIRE(rel && out_idx, "pg_dpp_ex: NULL argument");
    PG_REQUIRE(o->norm_relevance_score >= 0 && o->norm_relevance_score <= 2, "pg_dpp_ex: norm_relevance_score must be 0, 1 or 2");
    PG_REQUIRE(!o->has_table || (t && cand_rows), "pg_dpp_ex: has_table needs a table and candidate rows");
    PG_REQUIRE(o->has_table || o->hook_dim > 0, "pg_dpp_ex: no embedding table and no hook embeddings (the reference returns the items unchanged)");
    PG_REQUIRE(o->hook_dim == 0 || hook_emb, "pg_dpp_ex: hook_dim > 0 but hook_emb is NULL");
    const uint32_t dim = o->has_table ? t->dim : 0u;
    if (n > 8192 || o->hook_dim + dim > 4096) {
        pg::set_error("pg_dpp: %u candidates x %u dims unsupported (<= 8192 x 4096; the reference caps N with CandidateCount)", n, o->hook_dim + dim);
        return PG_ERR_UNSUPPORTED;
    }
    if (o->has_table)
        for (uint32_t i = 0; i < n; ++i)
            PG_REQUIRE(cand_rows[i] < t->rows, "pg_dpp: candidate row %u outside table", cand_rows[i]);
    // dpp_norm_relevance_score (dpp_sort.go:382-405): O(n) scalar work on the caller's thread, in the reference's
    // operation order (stat.PopMeanVariance two-pass with compensation, stat.StdScore; min-max takes max = first,
    // min = last item: the candidates arrive sorted by score)
    std::vector<double> rs(n);
    if (!pg::dpp_norm_relevance_host(rel, n, o->norm_relevance_score, rs.data())) {
        pg::set_error("pg_dpp: all item score is zero (dpp_sort.go:385-397); the caller keeps the items unchanged");
        return PG_ERR_ARITH;
    }
    if (out_relevance) memcpy(out_relevance, rs.data(), (size_t)n * 8);      // "dpp_relevance_score" (:410)
    const uint32_t topn = o->topn;
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr;
    if (o->has_table) tr = pg::TableRead(t->rw);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t bCand = al((size_t)n * 4), bRel = al((size_t)n * 8), bEmb = al((size_t)n * std::max(dim, 1u) * 4);
    const size_t bHook = al((size_t)n * std::max(o->hook_dim, 1u) * 8), bOut = al((size_t)(topn + 1) * 4 + 16);
    void* buf;
    int rc;
    if ((rc = pg::scratch_reserve(ctx, 5, bCand + bRel + bEmb + bHook + bOut, &buf))) return rc;
    char* p = (char*)buf;
    uint32_t* d_cand = (uint32_t*)p; p += bCand;
    double* d_rel = (double*)p; p += bRel;
    float* d_emb = (float*)p; p += bEmb;
    double* d_hook = (double*)p; p += bHook;
    uint32_t* d_out = (uint32_t*)p;
    uint32_t* d_cnt = d_out + topn;
    PG_HIP(hipMemcpyAsync(d_rel, rs.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    if (o->has_table) {
        PG_HIP(hipMemcpyAsync(d_cand, cand_rows, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = pg::table_gather_locked(ctx, t, d_cand, n, d_emb))) return rc;
    }
    if (o->hook_dim) PG_HIP(hipMemcpyAsync(d_hook, hook_emb, (size_t)n * o->hook_dim * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::dpp_run_locked(ctx, o->has_table ? d_emb : nullptr, o->hook_dim ? d_hook : nullptr, d_rel, 1, n, dim, o->hook_dim,
                                 o->alpha, topn, o->window, o->normalize_emb, o->ensure_pos_similarity, o->has_table, d_out, d_cnt)))
        return rc;
    PG_HIP(hipMemcpyAsync(ctx->h_status + 330, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    const uint32_t cnt = ctx->h_status[330];
    PG_HIP(hipMemcpyAsync(out_idx, d_out, (size_t)cnt * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    *out_count = cnt;
    return PG_OK;
}

int pg_dpp(pg_ctx* ctx, const pg_table* t, const uint32_t* cand_rows, const double* rel, uint32_t n,
           double alpha, uint32_t topn, uint32_t window, int normalize_emb, uint32_t* out_idx,
           uint32_t* out_count) {
    PG_REQUIRE(ctx && t && out_count, "pg_dpp: NULL argument");
    pg_dpp_options o;
    memset(&o, 0, sizeof o);
    o.alpha = alpha;
    o.topn = topn;
    o.window = window;
    o.normalize_emb = normalize_emb;
    o.ensure_pos_similarity = 1;
    o.has_table = 1;
    return pg_dpp_ex(ctx, t, cand_rows, rel, n, &o, nullptr, out_idx, out_count, nullptr);
}

}  // extern "C"
