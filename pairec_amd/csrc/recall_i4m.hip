// recall_i4m.hip — the small- and mid-batch screen: the full table pass of a recall with 1 … kI4mMaxQueries queries streams the
// 4-bit shadow of the rows (recall_i4.hip: 68 B per row instead of the int8 shadow's 128) through the int8 matrix
// pipe, then thins its suspects on the int8 shadow before the exact re-scoring.  Such a pass is HBM-bound on the
// shadow it streams (MFMA busy 0.36 at 128 queries on the int8 shadow), so the bytes are the cost.  Reference path:
// the same VectorRecall.GetCandidateItems as recall.hip (service/recall/vector_recall.go:32-123; one call per request,
// service/recall.go:126-150 fans them out) — results are bit-identical, the screens only decide what is re-scored.
//
// Stage 1, screen4m_kernel.  Shadow and bound are recall_i4.hip's: x^ = s_r X, X in [-7, 7] stored as X + 8, one
// scale per row, R_r >= ||x - x^|| measured, H_r = min(7 sqrt(dim) s_r, N + R4) >= ||x^||; for the int8 query
// q^ = s_q Q and I = sum X_i Q_i a row can reach thr only if
//     s_r s_q I + R_r B_q + H_r A_q >= thr          B_q = ||q|| (1 + slack),  A_q = ||q - q^|| + slack ||q||
// (slack: the rounding of the specification's fmaf chain, <= 128 x 2^-24 ||x|| ||q||).  Divided by s_r s_q > 0:
//     I >= tau_q u_r - (rho_r beta_q + eta_r alpha_q),    u = 1 / s_r, rho = R_r u, eta = H_r u,
//     tau = thr / s_q, beta = B_q / s_q, alpha = A_q / s_q
// and with kappa >= alpha_q / beta_q for every query of the batch, rho beta + eta alpha <= (rho + eta kappa) beta =: m_r beta_q.
// The MFMA's roles are swapped against screen_kernel's: the QUERIES are the A operand (resident for the launch), the
// table rows the B operand, so a lane owns ONE row (both halves of the wave: lane & 31) and its 16 accumulators per
// query block are 16 queries.  Everything per row (u, m) is then per lane, everything per query (tau, beta) is a
// loop-invariant register, and the test of a (row, query) pair is two fused multiply-adds (packed, v_pk_fma_f32) and a
// compare:  t = MAGIC - 8 - m beta_q;  t2 = tau_q u + t;  suspect iff !(float(acc) < t2)  where the accumulators start
// at bits(MAGIC) - 8 sum(Q) (MAGIC = 1.5 x 2^23: the int32 accumulator READ AS A FLOAT is MAGIC + I exactly, |I| < 2^22;
// the 8 sum(Q) takes the +8 of the stored nibbles out again).  Rounding: t and t2 round to integers in [2^23, 2^24)
// (<= 0.5 each), u is v_rcp_f32 (1 ulp), tau one division: together < 6 units for |tau u| < 2^24, covered by the 8;
// m is inflated by 1e-6, beta by 2e-6, kappa by 1e-6 (their own roundings are < 4e-7).  Out of range (|tau u| >= 2^24:
// the margin m beta is < 2^14) the float compare decides by sign and magnitude, as it should; a NaN cannot arise from
// finite constants (tau = -inf, beta = 0 for a query whose constants are not finite: every row a suspect — the lists
// overflow and the next plan runs), and the compare is written so that a NaN would be a suspect too.
//
// Data movement: a PIECE = 64 rows = 4 KiB of nibbles + 256 B of {s_r, R_r} words; HBM → LDS by LDS-DMA (four
// global_load_lds_dwordx4 + one global_load_lds_dword per piece), wave-private ring of 4 pieces, counted
// s_waitcnt vmcnt.  The 16-B quads of a row are rotated by (row >> 2) & 3 on the DMA's SOURCE side, so that a lane's
// two ds_read_b128 (its half of the row: 64 nibbles = the B operand of four k-steps) are conflict-free in the
// instruction's four 16-lane groups.  Unpack: three VALU per packed dword (and / shift / and).
//
// Stage 2, rescreen8_kernel.  The 4-bit bound reaches ~0.8 (uniform rows) to ~1.3 (Gaussian) score spreads below the
// threshold: 0.2–0.5 % of the rows are suspects of a query, 40–100 x the answer.  Re-scoring them exactly would gather
// 512 B each (0.1–0.2 ms per query and pass — what limits recall_i4.hip's VALU kernel to four queries).  Instead each
// suspect's int8 shadow row (ONE 128-B line) is gathered and tested with screen_kernel's integer bound
// (I8 >= thr_screen[q], recall.hip) — the suspects that remain are those the int8 screen would have found,
// ~9 K per query at K = 5 000, and rescore_kernel scores them from the fp32 rows as ever.
#include "common.hpp"

namespace pg {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kMWaves = 8;                                   // two per SIMD
constexpr int kMSlots = 4;                                   // ring: 1 consumed + 3 in flight per wave
constexpr int kMPieceRows = 64;
constexpr int kMPieceNib = kMPieceRows * 64;                 // 4 KiB
constexpr int kMPieceBytes = kMPieceNib + kMPieceRows * 4;   // + the rows' {s_r, R_r} words
constexpr int kMDmas = 5;                                    // LDS-DMA instructions per piece
constexpr int kMRing = kMWaves * kMSlots * kMPieceBytes;     // 136 KiB
constexpr int kMLds = 163840;
constexpr int kMStageW = (kMLds - kMRing) / kMWaves;         // 3 KiB of suspect staging per wave
constexpr float kMagic = 12582912.0f;                        // 1.5 x 2^23
constexpr int kMagicBits = 0x4B400000;
constexpr uint32_t kQcAt = kI4mMaxQueries * 32;              // words: the per-query constants behind the plain int8 queries
constexpr uint32_t kKappaAt = kQcAt + kI4mMaxQueries * 4;

struct Screen4mArgs {
    const char* d4;           // [rows + 64][64 B] nibbles (pg_table::d4)
    const uint32_t* d4s;      // [rows + 64] s_r (bf16, low half) | R_r (bf16, high half)
    const uint32_t* q4m;      // [64][32] int8 queries in plain order | [64][4] {s_q, beta_q, 8 sum(Q) as int bits, -} | kappa
    const float* thr;         // [>= nq] running thresholds
    float h_cap;              // N + R4
    uint32_t* susp_cnt;
    uint32_t* susp;           // [nq][cap]
    uint32_t* overflow;
    uint32_t cap, nq, row_begin, row_end;      // rows [row_begin, row_end), row_begin a multiple of 64
};

// One LDS-DMA instruction: global → LDS without touching VGPRs (recall.hip's dma_one: M0 saved and restored around it, the
// immediate offset unused — on an LDS-DMA it is added to the LDS address too; `after` is a value returned by a ds_read of the
// piece being computed, so that every read of the slot this DMA overwrites has completed: LDS returns in order).
__device__ __forceinline__ void dma_x4(const char* base, uint32_t lds_addr, uint32_t voff, int after) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_addr), "v"(after)
        : "memory");
}
__device__ __forceinline__ void dma_x1(const char* base, uint32_t lds_addr, uint32_t voff, int after) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, %2 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_addr), "v"(after)
        : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// NQB query blocks of 32 (1: up to 32 queries, 2: 33..64)
template <int NQB>
__global__ __launch_bounds__(64 * kMWaves, 2) void screen4m_kernel(Screen4mArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = kMSlots;
    constexpr int kCap = (kMStageW - NQB * 256 - 16) / 8;    // staged (row, query) pairs per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i32 = lane & 31, h = lane >> 5;

    // ---- launch-invariant: the queries as A fragments, their constants per accumulator register
    // A fragment of k-step j: lane (m, h) holds Q[m][64 h + 16 j .. + 15] — the dims whose nibbles a lane of the same half
    // unpacks for k-step j below (packed dwords 8 h + 2 j, 8 h + 2 j + 1 of its row)
    i32x4 afrag[NQB][4];
    f32x2 tau[NQB][8], beta[NQB][8];
    i32x16 cinit[NQB];
#pragma unroll
    for (int c = 0; c < NQB; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            afrag[c][j] = *reinterpret_cast<const i32x4*>(a.q4m + (size_t)(c * 32 + i32) * 32 + 16 * h + 4 * j);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t qi = (uint32_t)(c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
            const float* qc = reinterpret_cast<const float*>(a.q4m + kQcAt) + qi * 4;
            const float sq = qc[0], b = qc[1];
            const float t = a.thr[qi < a.nq ? qi : 0];
            // a query whose constants are not finite (beta = NaN from the prep kernel) or whose threshold is still open: every row
            // is a suspect; a query column beyond the batch: none is
            const bool ok = b == b && t == t && t > -__builtin_inff();
            float tv = ok ? t / sq : -__builtin_inff();
            if (qi >= a.nq) tv = __builtin_inff();
            const float bv = ok ? b : 0.0f;
            if (r & 1) { tau[c][r >> 1].y = tv; beta[c][r >> 1].y = bv; }
            else { tau[c][r >> 1].x = tv; beta[c][r >> 1].x = bv; }
            cinit[c][r] = kMagicBits - __float_as_int(qc[2]);
        }
    }
    const float kappa = reinterpret_cast<const float*>(a.q4m)[kKappaAt];
#pragma unroll
    for (int c = 0; c < NQB; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(afrag[c][j].x), "+v"(afrag[c][j].y), "+v"(afrag[c][j].z), "+v"(afrag[c][j].w));
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(tau[c][r]), "+v"(beta[c][r]));
        // (element by element: an asm operand of a whole 64-byte vector silently drops the kernel's host stub)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(cinit[c][r]));
    }

    // ---- this wave's run of pieces
    const uint32_t npieces = (a.row_end - a.row_begin + kMPieceRows - 1) / kMPieceRows;
    const uint32_t W = gridDim.x * kMWaves;
    const uint32_t gw = blockIdx.x * kMWaves + wave;
    const uint32_t ppw = (npieces + W - 1) / W;
    const uint32_t first = gw * ppw;
    const uint32_t np = first < npieces ? (npieces - first < ppw ? npieces - first : ppw) : 0;
    if (np == 0) return;

    // DMA lane offsets: LDS quad s = 64 n + lane of a piece is (row i = s >> 2, position p = s & 3) and receives the row's
    // quad (p - (i >> 2)) & 3, so that quad c of row i sits at position (c + (i >> 2)) & 3
    uint32_t voff[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int s = n * 64 + lane, i = s >> 2, p = s & 3;
        voff[n] = (uint32_t)(i * 64 + 16 * ((p - (i >> 2)) & 3));
    }
    const uint32_t voff_s = (uint32_t)lane * 4;
    const uint32_t lds_wave_u = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + wave * (NS * kMPieceBytes));
    char* const lds_ptr = smem + wave * (NS * kMPieceBytes);
    // this lane's reads inside a piece: row 32 k + i32 of block k, quads 2 h and 2 h + 1
    int rd[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = 32 * k + i32;
#pragma unroll
        for (int e = 0; e < 2; ++e) rd[k][e] = i * 64 + 16 * ((2 * h + e + (i >> 2)) & 3);
    }
    auto issue = [&](uint32_t rel, int after) {              // piece `rel` of the run (clamped to its last) into its ring slot
        const uint32_t pc = first + (rel < np ? rel : np - 1);
        const uint64_t row0 = (uint64_t)a.row_begin + (uint64_t)pc * kMPieceRows;
        const char* src = a.d4 + row0 * 64;
        const char* src_s = reinterpret_cast<const char*>(a.d4s + row0);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)(uintptr_t)src >> 32));
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)src);
        const uint32_t hi_s = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)(uintptr_t)src_s >> 32));
        const uint32_t lo_s = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)src_s);
        const char* ub = (const char*)(((uint64_t)hi << 32) | lo);
        const char* ub_s = (const char*)(((uint64_t)hi_s << 32) | lo_s);
        const uint32_t dst = lds_wave_u + __builtin_amdgcn_readfirstlane(rel % NS) * kMPieceBytes;
#pragma unroll
        for (int n = 0; n < 4; ++n) dma_x4(ub, dst + n * 1024, voff[n], after);
        dma_x1(ub_s, dst + kMPieceNib, voff_s, after);
    };

    // staging of (row, query) suspects, flushed in bulk into the per-query lists (screen_kernel's scheme)
    uint32_t* const st_row = reinterpret_cast<uint32_t*>(smem + kMRing + wave * kMStageW);
    uint32_t* const st_q = st_row + kCap;
    uint32_t* const st_cnt = st_q + kCap;                 // [NQB * 32]
    uint32_t* const st_base = st_cnt + NQB * 32;          // [NQB * 32]
    uint32_t st_n = 0;
    auto flush = [&]() {
        if (lane < NQB * 32) st_cnt[lane] = 0;
        for (uint32_t e0 = 0; e0 < st_n; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < st_n) atomicAdd(&st_cnt[st_q[e]], 1u);
        }
        if (lane < NQB * 32) {
            const uint32_t c = st_cnt[lane];
            st_base[lane] = c ? atomicAdd(&a.susp_cnt[lane], c) : 0u;
            st_cnt[lane] = 0;
        }
        for (uint32_t e0 = 0; e0 < st_n; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < st_n) {
                const uint32_t q = st_q[e];
                const uint32_t pos = st_base[q] + atomicAdd(&st_cnt[q], 1u);
                if (pos < a.cap) a.susp[(uint64_t)q * a.cap + pos] = st_row[e];
                else *a.overflow = 1u;
            }
        }
        st_n = 0;
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the stores must not disturb the ring's counted waits
    };

#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue((uint32_t)t, 0);

    const float magic8 = kMagic - 8.0f;
    for (uint32_t p = 0; p < np; ++p) {
        wait_vm<kMDmas * (NS - 2)>();
        const char* slot = lds_ptr + (p % NS) * kMPieceBytes;
        i32x4 pk[2][2];
        uint32_t srw[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            pk[k][0] = *reinterpret_cast<const i32x4*>(slot + rd[k][0]);
            pk[k][1] = *reinterpret_cast<const i32x4*>(slot + rd[k][1]);
            srw[k] = *reinterpret_cast<const uint32_t*>(slot + kMPieceNib + 4 * (32 * k + i32));
        }
        __builtin_amdgcn_sched_barrier(0);
        issue(p + NS - 1, (int)srw[1]);
        const uint32_t prow = a.row_begin + (first + p) * kMPieceRows;
        // both blocks' MFMAs first: the test of block 0 (vector ALU) then runs while block 1's are still in the matrix pipe — with the
        // blocks one after the other a wave's own MFMAs and tests never overlapped (64 queries: MFMA 0.25 + VALU ~0.6 of the time, in series)
        i32x16 acc[2][NQB];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int w8[8] = {pk[k][0].x, pk[k][0].y, pk[k][0].z, pk[k][0].w, pk[k][1].x, pk[k][1].y, pk[k][1].z, pk[k][1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w0 = (uint32_t)w8[2 * j], w1 = (uint32_t)w8[2 * j + 1];
                const i32x4 b = {(int)(w0 & 0x0F0F0F0Fu), (int)((w0 >> 4) & 0x0F0F0F0Fu), (int)(w1 & 0x0F0F0F0Fu),
                                 (int)((w1 >> 4) & 0x0F0F0F0Fu)};
#pragma unroll
                for (int c = 0; c < NQB; ++c)
                    acc[k][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[c][j], b, j == 0 ? cinit[c] : acc[k][c], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t row = prow + 32 * k + i32;
            // per row: u = 1 / s_r, m = (R_r + H_r kappa) u, inflated
            const float S = __uint_as_float(srw[k] << 16), R = __uint_as_float(srw[k] & 0xffff0000u);
            const float u = __builtin_amdgcn_rcpf(S);
            const float eta = fminf(79.1962f, a.h_cap * u);
            const float m = __fmaf_rn(eta, kappa, R * u) * 1.000001f;
            // the test: bit (16 NQB - 1 - (16 c + r)) of m32 <-> accumulator r of query block c
            const f32x2 nm2 = {-m, -m}, u2 = {u, u}, mg2 = {magic8, magic8};
            uint32_t m32 = 0;
#pragma unroll
            for (int c = 0; c < NQB; ++c)
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    const f32x2 t = __builtin_elementwise_fma(nm2, beta[c][r2], mg2);
                    const f32x2 t2 = __builtin_elementwise_fma(tau[c][r2], u2, t);
                    // m32 = 2 m32 + hit: the compare's lane mask (compiler-visible: it pads the MFMA -> VALU hazard of the accumulators)
                    // shifted in as the carry of an add — one instruction instead of a select and a shift-or
                    const uint64_t k0 = __builtin_amdgcn_fcmpf(__int_as_float(acc[k][c][2 * r2]), t2.x, 11);          // 11: unordered or >=
                    asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(m32) : "s"(k0) : "vcc");
                    const uint64_t k1 = __builtin_amdgcn_fcmpf(__int_as_float(acc[k][c][2 * r2 + 1]), t2.y, 11);
                    asm volatile("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(m32) : "s"(k1) : "vcc");
                }
            if (row >= a.row_end) m32 = 0;
            // stage the hits, one per lane per round (ballot + prefix count)
            for (;;) {
                const bool pnd = m32 != 0;
                const uint64_t bm = __builtin_amdgcn_ballot_w64(pnd);
                if (bm == 0) break;
                const uint32_t n = (uint32_t)__popcll(bm);
                if (st_n + n > (uint32_t)kCap) flush();
                if (pnd) {
                    const int s = 16 * NQB - 1 - __builtin_ctz(m32);
                    const int r = s & 15;
                    const uint32_t pos = st_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                    st_row[pos] = row;
                    st_q[pos] = (uint32_t)((s >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
                    m32 &= m32 - 1;
                }
                st_n += n;
            }
        }
    }
    flush();
    wait_vm<0>();
}

// per call: the int8 queries (screen_prep8_kernel's quantisation: the same Q and s_q the int8 stage's thresholds mean) in
// plain order, the bound's per-query constants and kappa.  One workgroup, four threads per query.
__global__ __launch_bounds__(256) void screen4m_prep_kernel(const float* __restrict__ qpad, uint32_t nq, uint32_t* __restrict__ q4m,
                                                            uint32_t* __restrict__ susp2_cnt) {
    __shared__ float s_ratio[64];
    const uint32_t tid = threadIdx.x, qi = tid >> 2, part = tid & 3;
    if (tid <= kI4mMaxQueries) susp2_cnt[tid] = 0u;         // the int8 stage's counters + its statistics word, for the job's first launch
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
    const float sc = fmaxf(mx / 127.0f, 1e-30f);
    double ss = 0.0, dd = 0.0;
    int sum = 0;
    uint32_t* out = q4m + (size_t)qi * 32 + part * 8;
    for (int e = 0; e < 8; ++e) {
        uint32_t word = 0;
        for (int b = 0; b < 4; ++b) {
            const float f = q[4 * e + b];
            int Q = __float2int_rn(f / sc);
            Q = Q > 127 ? 127 : (Q < -127 ? -127 : Q);
            if (bad) Q = 0;
            const double v = (double)f, d = v - (double)sc * (double)Q;
            ss += v * v;
            dd += d * d;
            sum += Q;
            word |= (uint32_t)(Q & 0xff) << (8 * b);
        }
        out[e] = word;
    }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    dd += __shfl_xor(dd, 1, 64);
    dd += __shfl_xor(dd, 2, 64);
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    if (part == 0) {
        const double nrm = sqrt(ss), dq = sqrt(dd);
        // B: the factor of the row's residual; A: of its norm bound (1e-5: the specification's fp32 chain rounds by
        // <= 128 x 2^-24 ||x|| ||q|| <= 1e-5 (H_r + R_r) ||q||; the sums of squares were accumulated in double)
        const double B = nrm * (1.0 + 1e-5) * 1.000001 + 1e-30;
        const double A = dq * 1.0001 + 1e-5 * nrm + 1e-30;
        const double beta = B / (double)sc * (1.0 + 2e-6), ratio = A / B;
        const bool fin = !bad && beta == beta && beta < 1e30 && ratio == ratio && ratio < 1e30;
        float* c = reinterpret_cast<float*>(q4m + kQcAt) + qi * 4;
        c[0] = sc;
        c[1] = fin ? (float)beta : __builtin_nanf("");
        c[2] = __int_as_float(8 * sum);
        c[3] = 0.0f;
        s_ratio[qi] = (fin && qi < nq) ? (float)(ratio * (1.0 + 1e-6)) : 0.0f;
    }
    __syncthreads();
    if (tid == 0) {
        float k = 0.0f;
        for (uint32_t i = 0; i < kI4mMaxQueries; ++i) k = fmaxf(k, s_ratio[i]);
        reinterpret_cast<float*>(q4m)[kKappaAt] = k * 1.000001f;
    }
}

// Stage 2: the stage-1 suspects of every query against the int8 shadow.  A wave walks chunks of 64 suspects of one query:
// eight lanes gather a suspect's 128-B row (sixteen bytes each: one full line per suspect), four v_dot4_i32_i8 per lane, and the
// eight rows' eight partial sums of a lane group are reduced ACROSS the group in seven exchanges (a transposing butterfly: lane
// 8 g + i ends with the total of the group's row i — its own suspect).  Survivors (I8 >= the int8 screen's integer threshold)
// are compacted into the list rescore_kernel reads.  The gathers of the next chunk are in flight under the arithmetic of this
// one and the suspect rows are read two chunks ahead: random 128-B lines reach 6.4 TB/s on this part when enough of them are
// requested (scripts/micro/gather128.hip), a wave that waits for one chunk at a time gets 1.8.  grid (blocks, nq), 256 threads.
constexpr int kS2Stage = 448;            // staged survivors per wave (flushed above 384)
__device__ __forceinline__ void rs8_gather(i32x4 (&v)[8], const int8_t* __restrict__ d8, uint32_t row, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r_i = (uint32_t)__shfl((int)row, (lane & 56) + i, 64);
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(d8 + (size_t)r_i * 128) + (lane & 7));
    }
}
__device__ __forceinline__ int rs8_reduce(const i32x4 (&v)[8], const i32x4& qd, int lane) {
    int p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int acc = __builtin_amdgcn_sdot4(v[i].x, qd.x, 0, false);
        acc = __builtin_amdgcn_sdot4(v[i].y, qd.y, acc, false);
        acc = __builtin_amdgcn_sdot4(v[i].z, qd.z, acc, false);
        p[i] = __builtin_amdgcn_sdot4(v[i].w, qd.w, acc, false);
    }
    // lane bit 2 keeps rows 4..7 (else 0..3) and hands the other half to its partner; then bit 1, then bit 0
    const bool b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    int s4[4], s2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) s4[i] = (b2 ? p[i + 4] : p[i]) + __shfl_xor(b2 ? p[i] : p[i + 4], 4, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) s2[i] = (b1 ? s4[i + 2] : s4[i]) + __shfl_xor(b1 ? s4[i] : s4[i + 2], 2, 64);
    return (b0 ? s2[1] : s2[0]) + __shfl_xor(b0 ? s2[0] : s2[1], 1, 64);
}
__global__ __launch_bounds__(256) void rescreen8_kernel(const int8_t* __restrict__ d8, const uint32_t* __restrict__ q4m,
                                                        const float* __restrict__ thr_screen, const uint32_t* __restrict__ susp,
                                                        const uint32_t* __restrict__ susp_cnt, uint32_t scap, uint32_t n_rows,
                                                        uint32_t* __restrict__ susp2, uint32_t* __restrict__ susp2_cnt, uint32_t cap2,
                                                        uint32_t* __restrict__ overflow, uint32_t* __restrict__ stat) {
    __shared__ uint32_t stage[4][kS2Stage];
    const uint32_t q = blockIdx.y;
    const uint32_t n_raw = susp_cnt[q];
    const uint32_t n = n_raw < scap ? n_raw : scap;
    if (n_raw > scap && blockIdx.x == 0 && threadIdx.x == 0) *overflow = 1u;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(stat, n);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int T = __float_as_int(thr_screen[q]);
    const i32x4 qd = *reinterpret_cast<const i32x4*>(q4m + (size_t)q * 32 + (lane & 7) * 4);
    const uint32_t nchunks = (n + 63) / 64, cstep = gridDim.x * 4;
    uint32_t c = blockIdx.x * 4 + (uint32_t)w;
    if (c >= nchunks) return;
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
    // (a list that overflowed has holes whose stale contents may be rows of an earlier, larger table: stay inside this one)
    auto rows_of = [&](uint32_t ch) -> uint32_t {
        const uint32_t e = ch * 64 + lane;
        const uint32_t r = (ch < nchunks && e < n) ? susp[(uint64_t)q * scap + e] : 0u;
        return r < n_rows ? r : 0u;
    };
    auto emit = [&](uint32_t ch, uint32_t row, int mine) {
        const bool keep = ch * 64 + lane < n && mine >= T;
        const uint64_t bm = __builtin_amdgcn_ballot_w64(keep);
        if (bm) {
            const uint32_t k = (uint32_t)__popcll(bm);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (keep) stage[w][cnt + before] = row;
            cnt += k;
            if (cnt > (uint32_t)(kS2Stage - 64)) flush();
        }
    };
    i32x4 va[8], vb[8];
    uint32_t rowA = rows_of(c), rowB = rows_of(c + cstep);
    rs8_gather(va, d8, rowA, lane);
    for (;;) {
        // chunk c's rows are in flight in va; rowB holds the rows of chunk c + cstep
        const bool hasB = c + cstep < nchunks;
        if (hasB) rs8_gather(vb, d8, rowB, lane);
        const uint32_t rowC = rows_of(c + 2 * cstep);
        emit(c, rowA, rs8_reduce(va, qd, lane));
        if (!hasB) break;
        c += cstep;
        const bool hasC = c + cstep < nchunks;
        if (hasC) rs8_gather(va, d8, rowC, lane);
        const uint32_t rowD = rows_of(c + 2 * cstep);
        emit(c, rowB, rs8_reduce(vb, qd, lane));
        if (!hasC) break;
        c += cstep;
        rowA = rowC;
        rowB = rowD;
    }
    if (cnt) flush();
}

}  // namespace

int screen4m_prep_launch(pg_ctx* ctx, const RecallScratch& rs, uint32_t nq) {
    screen4m_prep_kernel<<<1, 256, 0, ctx->stream>>>(rs.qpad, nq, rs.q4m, rs.susp2_cnt);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// stage 1 over rows [row_begin, row_end) (row_begin a multiple of 64) into rs.susp as [nq][cap1], then stage 2 into
// rs.susp2 as [nq][rs.cap]; rs.susp_cnt must be zero on entry, rs.susp2_cnt is zeroed here unless the caller says it still is
// (the job's first launch: screen4m_prep_kernel left it so)
int screen4m_launch(pg_ctx* ctx, const pg_table* t, const RecallScratch& rs, uint32_t nq, uint32_t row_begin, uint32_t row_end,
                    uint32_t cap1, bool susp2_clean) {
    Screen4mArgs a;
    a.d4 = reinterpret_cast<const char*>(t->d4);
    a.d4s = t->d4s;
    a.q4m = rs.q4m;
    a.thr = rs.thr;
    a.h_cap = (t->max_norm + t->rmax4) * 1.000001f;
    a.susp_cnt = rs.susp_cnt;
    a.susp = rs.susp;
    a.overflow = rs.overflow;
    a.cap = cap1;
    a.nq = nq;
    a.row_begin = row_begin;
    a.row_end = row_end;
    const uint32_t npieces = (row_end - row_begin + kMPieceRows - 1) / kMPieceRows;
    uint32_t grid = (uint32_t)ctx->num_cus;
    if (grid > (npieces + kMWaves - 1) / kMWaves) grid = (npieces + kMWaves - 1) / kMWaves;
    if (grid == 0) grid = 1;
    int rc;
    if (!susp2_clean) PG_HIP(hipMemsetAsync(rs.susp2_cnt, 0, sizeof(uint32_t) * (kI4mMaxQueries + 1), ctx->stream));
    uint32_t* const stat = rs.susp2_cnt + kI4mMaxQueries;      // (zeroed with the counters; copied out with the job's status words)
    if (nq <= 32) {
        if ((rc = ensure_dyn_lds(ctx, (const void*)screen4m_kernel<1>, kMLds))) return rc;
        screen4m_kernel<1><<<grid, 64 * kMWaves, kMLds, ctx->stream>>>(a);
    } else {
        if ((rc = ensure_dyn_lds(ctx, (const void*)screen4m_kernel<2>, kMLds))) return rc;
        screen4m_kernel<2><<<grid, 64 * kMWaves, kMLds, ctx->stream>>>(a);
    }
    PG_HIP(hipGetLastError());
    // a wave should walk ~10 chunks of 64 suspects (its gathers run a chunk ahead: a wave with two or three chunks spends its time
    // on the first round trip — 8 queries: 104 us with 256 blocks per query, the 4-bit stage's pass-through known from earlier
    // batches puts it at a quarter of that); never more than the chip holds at once
    const double pairs = t->i4m_pairs > 0.0f ? (double)t->i4m_pairs : 170000.0;
    uint32_t s2_blocks = (uint32_t)(pairs / 64.0 / 10.0 / 4.0) + 1;
    if (s2_blocks > 2048u / nq) s2_blocks = 2048u / nq;
    if (s2_blocks < 8) s2_blocks = 8;
    rescreen8_kernel<<<dim3(s2_blocks, nq), 256, 0, ctx->stream>>>(t->d8, rs.q4m, rs.thr_screen, rs.susp, rs.susp_cnt, cap1, (uint32_t)t->rows,
                                                             rs.susp2, rs.susp2_cnt, rs.cap, rs.overflow, stat);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // namespace pg
