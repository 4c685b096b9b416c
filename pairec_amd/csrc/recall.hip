// recall.hip — exact inner-product top-K over an HBM-resident fp32 table.
//
// Replaces the reference's remote candidate generation: FaissModel.Run → VectorClient.Search
// (algorithm/faiss/model.go:29-31, vector_client.go:32-41) and Hologres'
// pm_approx_inner_product_distance ... ORDER BY distance desc LIMIT n
// (service/recall/hologres_vector_recall.go:23); call site service/recall/vector_recall.go:88-102.
//
// Specification:  score(row,q) = chain_{k asc} fmaf(x[row][k], q[k], acc), fp32 —
// exactly what gfx950's v_mfma_f32_32x32x2_f32 computes (a k-ordered fmaf chain, one rounding per
// step), so a request's scores do not depend on how many other requests share the table pass.
// Order: score descending (IEEE totalOrder, NaN last), then row ascending, via a 64-bit key.
//
// Kernels
//   scan_kernel      HBM-bound.  One wave = one 32-row block at a time; rows stream HBM → LDS by
//                    LDS-DMA (global_load_lds_dwordx4, full 128-B lines, no VGPR staging) through a
//                    per-wave ring of 8 KiB pieces (no workgroup barriers); A fragments are read
//                    from LDS conflict-free thanks to a per-row rotation applied on the DMA's
//                    *source* address; 32 queries ride in the B operand.  Rows whose score reaches
//                    the query's running threshold are appended to a candidate list.
//   screen_kernel    the same walk over a quantised shadow of the rows (int8 at dim 128, bf16 at dim 64): an
//                    int8 / bf16 MFMA bounds every row·query score rigorously, pairs whose bound reaches the
//                    threshold are "suspects"; rescore_kernel scores the suspects exactly from the fp32 rows.
//                    Up to 256 queries per pass; results identical to scan_kernel's, bit for bit.
//   select_kernel    per query: radix-select the K-th largest key of the candidates, keep the top
//                    K, publish the new threshold (any K-th-largest-so-far is a valid lower bound,
//                    so the result is exact for every data distribution).
//   final_kernel     per query: bitonic sort of the K survivors in LDS, decode to (row, score).
#include "common.hpp"
#include "bitonic_reg.hpp"
#include "split_sort.hpp"

#include <type_traits>

namespace pg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPieceRows = 32;
constexpr int kPieceCols = 32;                             // one 128-B line per row
constexpr int kPieceBytes = kPieceRows * kPieceCols * 4;   // 4 KiB
constexpr int kPieceDmas = kPieceBytes / 1024;             // LDS-DMA instructions per piece
constexpr int kPieceQuads = kPieceCols / 4;                // 16-B quads per row of a piece
constexpr int kRingSlots = 4;                              // per wave: 1 consumed + 3 in flight
constexpr int kScanWaves = 8;                              // waves per workgroup (2 per SIMD: one wave's
                                                           // waits/epilogues hide behind the other's MFMAs)
constexpr int kScanLdsRing = kScanWaves * kRingSlots * kPieceBytes;   // 128 KiB
constexpr int kStageCap = 280;                             // staged hits per wave (fills the LDS left by the ring)
constexpr int kStageBytes = kStageCap * 12 + 512;          // keys + query ids + 64 counters + 64 bases
constexpr int kScanLds = kScanLdsRing + kScanWaves * kStageBytes;

__device__ __forceinline__ uint32_t f32_ordered_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    if (f != f) return 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ uint64_t topk_key(float score, uint32_t row) {
    return ((uint64_t)f32_ordered_bits(score) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float key_score(uint64_t key) {
    const uint32_t ob = (uint32_t)(key >> 32);
    if (ob == 0u) return __uint_as_float(0x7FC00000u);
    const uint32_t b = (ob & 0x80000000u) ? (ob & 0x7FFFFFFFu) : ~ob;
    return __uint_as_float(b);
}
__device__ __forceinline__ uint32_t key_row(uint64_t key) {
    return 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
}

struct ScanArgs {
    const float* tab;        // [rows][DIM]
    const float* qpad;       // [32][DIM] queries, zero-padded
    const float* thr;        // [32] running thresholds
    uint32_t* cnt;           // [32] candidate counts
    uint64_t* cand;          // [32][cap] candidate keys
    uint32_t* overflow;      // set to 1 if any list overflowed
    uint32_t cap;
    uint32_t nq;
    uint32_t nq_launch;      // queries of the call (selects the 32- or 64-query kernel)
    uint32_t rb_begin;       // first 32-row block of this launch
    uint32_t rb_end;         // one past the last block
    uint32_t row_end;        // rows >= row_end are ignored (table end)
    uint32_t stride;         // 1 = every block; S > 1 = pilot sample: logical block L reads table
                             // block sample_block(L, ...)
    uint32_t perm_mul, perm_mod;   // pilot sample order: slot = L * perm_mul mod perm_mod
    uint32_t group_q;        // > 0: blockIdx.y selects a group of group_q queries (of nq in all) — one launch serves
                             // every group of a screened recall's exact seed scan instead of one launch per group
    // squared-Euclidean recall (scan_kernel<…, L2 = true>): a row·query pair is ranked by -d = fmaf(2, ip, -(|x|^2 + |q|^2))
    const float* nx;         // [rows + 64] |x|^2 of every row (k-ascending fmaf chain; pg_table::d_nx)
    const float* nqv;        // [nq] |q|^2 of every query
    RowFilter filter;        // rows that fail it are no candidates (col = nullptr: none)
};

// Pilot sample: logical block L of a stride-S launch is table block slot*S + jitter(slot), where
// slot = L*mul mod n_slots is a fixed permutation of the n_slots sample slots (mul coprime with
// n_slots) and jitter < S.  The permutation makes the pilot's own streaming order independent of the
// table order (a table sorted by score would otherwise defeat its running threshold); the jitter keeps
// a periodic layout from aliasing with the sample.  S = 1 is the identity.
__device__ __forceinline__ uint32_t slot_block(uint32_t slot, uint32_t S) {
    return slot * S + (uint32_t)(((uint64_t)(slot * 2654435761u) * S) >> 32);       // jitter in [0, S)
}
__device__ __forceinline__ uint32_t sample_block(uint32_t L, uint32_t S, uint32_t mul, uint32_t mod) {
    if (S == 1) return L;
    return slot_block((uint32_t)(((uint64_t)L * mul) % mod), S);
}

// One LDS-DMA instruction (64 lanes x 16 B = 1 KiB): global → LDS without touching VGPRs.
// `base` is the wave-uniform byte address of the piece (first row, first column of the piece),
// `voff` the lane's byte offset, `lds_addr` the wave-uniform LDS destination (the hardware adds
// lane*16).  M0 carries the LDS address; it is saved/restored because hipcc owns it outside this
// asm.  (The instruction's immediate offset is NOT used: on an LDS-DMA it is added to the LDS
// address too.)  `nt`: the table is streamed once, keep it out of the way of L2-resident data.
//
// WAR guard: `after` must be a value returned by a ds_read of the piece CURRENTLY being computed.
// It is an (unused) input of the asm, so hipcc waits for that read before the DMA issues; LDS
// returns in order, hence every read of the previous piece — whose slot this DMA overwrites — has
// completed too.  Without it a DMA served from L2/MALL was observed to land before a still-queued
// ds_read of the old piece (1 lost candidate in ~10 % of small-table runs).
__device__ __forceinline__ void dma_one(const char* base, uint32_t lds_addr, uint32_t voff, float after) {
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// VAR: developer ablations (0 = product; 1 = no threshold test; 2 = no MFMA; 3 = no DMA)
// NQB = number of 32-query column blocks riding in the B operand (1: up to 32 queries, 2: up to 64).
// With two blocks every A fragment feeds two independent accumulator chains and the kernel becomes
// fp32-MFMA-bound instead of HBM-bound (2 x 64 cycles per 32 B of table per SIMD).
// L2: the candidates are the rows of SMALLEST squared Euclidean distance (service/recall/hologres_vector_recall_v2.go:23).  The
// inner products come out of the same MFMA chains; every accumulator is then turned into -d = fmaf(2, ip, -(|x|^2 + |q|^2))
// — the block's 32 row norms arrive by scalar loads (the block's first row is wave-uniform; lgkmcnt, not the ring's vmcnt) —
// and everything behind (threshold streaming, keys, select, final) works on -d unchanged.  HBM-bound like the exact scan.
template <int DIM, int NQB = 1, int VAR = 0, bool L2 = false>
__global__ __launch_bounds__(64 * kScanWaves, 2) void scan_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PPB = DIM / kPieceCols;       // pieces per 32-row block
    constexpr int NS = kRingSlots;
    constexpr int ND = kPieceDmas;              // 4
    constexpr int NQ = kPieceQuads;             // 8
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * kScanWaves + wave;
    const uint32_t W = gridDim.x * kScanWaves;
    const int i32 = lane & 31;       // row within block (A), query (B, C)
    const int h = lane >> 5;         // k parity (A, B); row half (C)
    if (a.group_q) {                 // this workgroup's query group
        const uint32_t g0 = blockIdx.y * a.group_q;
        a.qpad += (size_t)g0 * DIM;
        a.thr += g0;
        a.cnt += g0;
        a.cand += (uint64_t)g0 * a.cap;
        if (L2) a.nqv += g0;
        a.nq = a.nq - g0 < a.group_q ? a.nq - g0 : a.group_q;
    }

    // B operand: bq[s] = Q[query = lane&31][k = 2s + h]
    float bq[NQB][DIM / 2];
    float thr[NQB];
    float nq2[NQB];                  // L2: |q|^2 of this lane's query
    bool active[NQB];
#pragma unroll
    for (int c = 0; c < NQB; ++c) {
#pragma unroll
        for (int s = 0; s < DIM / 2; ++s) bq[c][s] = a.qpad[(c * 32 + i32) * DIM + 2 * s + h];
        thr[c] = a.thr[c * 32 + i32];
        active[c] = (uint32_t)(c * 32 + i32) < a.nq;
        nq2[c] = (L2 && active[c]) ? a.nqv[c * 32 + i32] : 0.0f;
        if (L2) asm volatile("" : "+v"(nq2[c]));
    }
    // Pin the operand loads' completion HERE: hipcc places a load's s_waitcnt at its first use,
    // which would otherwise land inside the streaming loop as vmcnt(0) and drain the DMA ring.
#pragma unroll
    for (int c = 0; c < NQB; ++c) {
#pragma unroll
        for (int s = 0; s < DIM / 2; ++s) asm volatile("" : "+v"(bq[c][s]));
        asm volatile("" : "+v"(thr[c]));
    }

    // DMA lane offsets.  A piece is 32 rows x 32 columns (one 128-B line per row); DMA n covers
    // LDS quad slots S = n*64 + lane ↔ (row i = S/8, quad p = S%8), which receive global quad
    // (p + i/2) % 8 of that row: the rotation makes the A-fragment ds_read_b128 conflict-free.
    uint32_t voff[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) {
        const int S = n * 64 + lane;
        const int i = S >> 3, p = S & 7;
        voff[n] = (uint32_t)(i * DIM + 4 * ((p + (i >> 1)) & 7)) * 4u;
    }
    const uint32_t lds_wave_u = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + wave * (NS * kPieceBytes));
    char* const lds_ptr = smem + wave * (NS * kPieceBytes);
    const int rd_row = i32 * 128;
    const int rd_rot = (i32 >> 1) * 16;

    // each wave walks a CONTIGUOUS run of row blocks (sequential 16 KiB reads), runs are spread
    // evenly over the launch's waves
    const uint32_t total = a.rb_end - a.rb_begin;
    const uint32_t bpw = (total + W - 1) / W;
    const uint32_t first = gw * bpw;
    const uint32_t nblk = first < total ? (total - first < bpw ? total - first : bpw) : 0;
    if (nblk == 0) return;

    // wave-uniform source base and LDS destination of piece t (tail pieces re-read the last block)
    auto piece_addr = [&](uint32_t t, const char*& ub, uint32_t& dst) {
        uint32_t b = t / PPB;
        if (b >= nblk) b = nblk - 1;
        const uint64_t rb = sample_block(a.rb_begin + first + b, a.stride, a.perm_mul, a.perm_mod);
        const char* base = (const char*)a.tab + rb * (uint64_t)(kPieceRows * DIM * 4) + (t % PPB) * (kPieceCols * 4);
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)(uintptr_t)base >> 32));
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
        ub = (const char*)(((uint64_t)hi << 32) | lo);
        dst = lds_wave_u + __builtin_amdgcn_readfirstlane(t % NS) * kPieceBytes;
    };

    // wave-private staging list for threshold hits (keys + query ids)
    uint64_t* const st_key = reinterpret_cast<uint64_t*>(smem + kScanLdsRing + wave * kStageBytes);
    uint32_t* const st_q = reinterpret_cast<uint32_t*>(st_key + kStageCap);
    uint32_t* const st_cnt = st_q + kStageCap;            // [64] per-query counts / running offsets
    uint32_t* const st_base = st_cnt + 64;                // [64] reserved base per query
    uint32_t st_n = 0;
    // Flush: reserve space per QUERY (<= 64 returning global atomics per flush instead of one per
    // hit — the list counters are the hottest words of the launch), then scatter.
    auto flush = [&]() {
        st_cnt[lane] = 0;
        for (uint32_t e0 = 0; e0 < st_n; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < st_n) atomicAdd(&st_cnt[st_q[e]], 1u);
        }
        {
            const uint32_t c = st_cnt[lane];
            st_base[lane] = c ? atomicAdd(&a.cnt[lane], c) : 0u;
            st_cnt[lane] = 0;
        }
        for (uint32_t e0 = 0; e0 < st_n; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < st_n) {
                const uint32_t q = st_q[e];
                const uint32_t pos = st_base[q] + atomicAdd(&st_cnt[q], 1u);
                if (pos < a.cap) a.cand[(uint64_t)q * a.cap + pos] = st_key[e];
                else *a.overflow = 1u;
            }
        }
        st_n = 0;
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), visible to hipcc's bookkeeping
    };

    // prologue: pieces 0..NS-2 in flight
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        const char* src;
        uint32_t dst;
        piece_addr(t, src, dst);
#pragma unroll
        for (int n = 0; n < ND; ++n) dma_one(src, dst + n * 1024, voff[n], 0.0f);
    }

    for (uint32_t b = 0; b < nblk; ++b) {
        f32x16 acc[NQB];
#pragma unroll
        for (int c = 0; c < NQB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
#pragma unroll
        for (int pc = 0; pc < PPB; ++pc) {
            const uint32_t t = b * PPB + pc;
            // pieces t..t+NS-2 are in flight (ND DMAs each); piece t+NS-1 is issued below, one DMA
            // every other quad, into the slot that piece t-1 just vacated
            if (VAR != 3) wait_vmcnt<ND * (NS - 2)>();      // piece t has landed
            const char* nb_src;
            uint32_t nb_dst;
            piece_addr(t + NS - 1, nb_src, nb_dst);
            const char* slot = lds_ptr + (t % NS) * kPieceBytes + rd_row;
            // all of the piece's A-fragment reads are issued up front (8 x ds_read_b128 in flight),
            // so LDS latency is paid once per piece instead of once per quad
            f32x4 qv[NQ];
#pragma unroll
            for (int g = 0; g < NQ; ++g)
                qv[g] = *reinterpret_cast<const f32x4*>(slot + ((g * 16 - rd_rot) & 112));
            __builtin_amdgcn_sched_barrier(0);      // keep the reads up here (hipcc would sink them to their uses)
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                const f32x4 q = qv[g];
                if (g < ND && VAR != 3) dma_one(nb_src, nb_dst + g * 1024, voff[g], q.x);
                const float a0 = h ? q.y : q.x;
                const float a1 = h ? q.w : q.z;
                const int s = pc * (kPieceCols / 2) + 2 * g;
                if (VAR == 2) {
                    acc[0][g] += a0 + a1;
                } else {
#pragma unroll
                    for (int c = 0; c < NQB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[c][s], acc[c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < NQB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[c][s + 1], acc[c], 0, 0, 0);
                }
            }
        }
        if (VAR == 1 || VAR == 3) {
#pragma unroll
            for (int c = 0; c < NQB; ++c) asm volatile("" ::"v"(acc[c][0]), "v"(acc[c][15]));   // keep the chain live
            continue;
        }
        if constexpr (L2) {
            const uint32_t l2_row0 = __builtin_amdgcn_readfirstlane(
                sample_block(a.rb_begin + first + b, a.stride, a.perm_mul, a.perm_mod) * kPieceRows);
            const float* const nxp = a.nx + l2_row0;               // wave-uniform: scalar loads
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const float nxa = nxp[ro], nxb = nxp[ro + 4];
                const float nxr = h ? nxb : nxa;
#pragma unroll
                for (int c = 0; c < NQB; ++c) acc[c][r] = __fmaf_rn(2.0f, acc[c][r], -(nxr + nq2[c]));
            }
        }
        // ---- threshold test: C layout col = lane&31 (query within its block), row = (r&3)+8*(r>>2)+4*h.
        // Fast path: one not-less-than compare per accumulator register, OR-reduced.
        bool any = false;
#pragma unroll
        for (int c = 0; c < NQB; ++c) {
            bool anyc = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) anyc |= !(acc[c][r] < thr[c]);
            any |= anyc && active[c];
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0) {
            // Hits are rare (≈ K/rows_seen per row·query), so they are parked in a wave-private LDS
            // staging list and flushed in bulk: a returning atomic costs a full vmcnt drain of the
            // DMA ring, which must not happen once per block.
            const uint32_t row0 = sample_block(a.rb_begin + first + b, a.stride, a.perm_mul, a.perm_mod) * kPieceRows;
#pragma unroll
            for (int c = 0; c < NQB; ++c) {
                const uint32_t qid = (uint32_t)(c * 32 + i32);
                uint32_t pass = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const bool p = active[c] && !(acc[c][r] < thr[c]) && row < a.row_end && row_filter_pass(a.filter, row < a.row_end ? row : 0u);
                    pass |= (p ? 1u : 0u) << r;
                }
                if (__builtin_amdgcn_ballot_w64(pass != 0) == 0) continue;
                uint32_t total_hits = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    total_hits += __popcll(__builtin_amdgcn_ballot_w64((pass >> r) & 1u));
                if (st_n + total_hits > (uint32_t)kStageCap) flush();
                if (total_hits > (uint32_t)kStageCap) {
                    // dense case (first chunk: threshold still -inf): straight to global memory
                    if (pass != 0) {
                        uint32_t pos = atomicAdd(&a.cnt[qid], (uint32_t)__popc(pass));
                        uint64_t* dst = a.cand + (uint64_t)qid * a.cap;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (pass & (1u << r)) {
                                const uint32_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                                if (pos < a.cap) dst[pos] = topk_key(acc[c][r], row);
                                else *a.overflow = 1u;
                                ++pos;
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), visible to hipcc's bookkeeping
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool p = (pass >> r) & 1u;
                        const uint64_t m = __builtin_amdgcn_ballot_w64(p);
                        if (m != 0) {
                            if (p) {
                                const uint32_t pos = st_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                                const uint32_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                                st_key[pos] = topk_key(acc[c][r], row);
                                st_q[pos] = qid;
                            }
                            st_n += __popcll(m);
                        }
                    }
                }
            }
        }
    }
    flush();
    wait_vmcnt<0>();   // drain the tail re-reads before the wave's LDS is released
}

// ---------------------------------------------------------------------------------------------
// Screened scan: the same stream, but the per-row scores are first *bounded* with bf16 MFMA and
// only rows that could beat the threshold are re-scored exactly.
//
//   s~ = sum_k bf16(x_k) * bf16(q_k)   (v_mfma_f32_32x32x16_bf16, 16x cheaper than the fp32 MFMA)
//   |s~ - s| <= eps_q = 0.008 * max_row_norm * ||q||          (s = the specification's fmaf chain)
//     [2*2^-8 + 2^-16 operand rounding (any rounding mode) + 2 * 128 * 2^-24 accumulation, times
//      sum|x_k q_k| <= ||x||*||q||]
// A row is staged when !(s~ < thr_q - eps_q); at flush time every staged (row, query) pair gets the
// exact k-ascending fmaf chain from the fp32 table and is kept only if !(s < thr_q).  Any member of
// the final top-K has s >= thr_q at every moment (thr is a lower bound of the final K-th score), so
// s~ >= thr_q - eps_q and it is staged: the result is identical, bit for bit, to the exact scan.
// With 128 queries (4 column blocks) per pass the MFMA work is 1/4 of the 32-query exact kernel's
// and the kernel stays HBM-bound.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int kScreenMaxNQB = 8;                // 8 x 32 = 256 queries per pass at most
constexpr int kScreenLds = 163840;              // the whole LDS of a CU: ring (128 KiB) + staging (32 KiB)
constexpr float kScreenEps = 0.008f;

struct ScreenArgs {
    const void* tab16;        // shadow of the table: bf16 (pg_table::d16) or int8 (pg_table::d8)
    const uint4* qb16;        // [NQB][KS][64] B fragments: 8 bf16 (KS = DIM/16) or 16 int8 (KS = DIM/32) per lane
    const float* thr_screen;  // [256] bf16: thr - eps, rounded down; int8: the same in integer dot units (int32 bits)
    uint32_t* susp_cnt;       // [256] suspects per query of this launch
    uint32_t* susp;           // [256][cap] suspect rows (passed the screen; re-scored by rescore_kernel)
    uint32_t* overflow;
    uint32_t cap, nq, rb_begin, rb_end, row_end, stride, perm_mul, perm_mod;
    // bf16 only: the bound is relative to the row's norm — eps = eps_unit[q] x (largest row norm of the 32-row block)
    const float* blk_norm2;   // [blocks] largest row norm^2 per physical 32-row block (pg_table::dnorm2)
    const float* eps_unit;    // [256] 0.008 ||q|| (inflated); thr_screen then carries thr itself (or -inf)
    // query halves (int8, > 128 queries): hit records instead of staged (row, query) pairs — see kRecBytes
    char* rec;                // [waves][rec_cap] records of kRecBytes: one region per wave of the launch
    uint32_t* rec_cnt;        // [waves] records in each region (zeroed per launch); [waves]: records in the spill pool
    uint32_t rec_cap;
    char* rec_pool;           // [rec_pool_cap] records: where a wave whose region is full puts its staged records
    uint32_t rec_pool_cap, rec_waves;
    uint32_t early_share;     // 8-wave variants: share (x 1024) of a SIMD's blocks that goes to its older wave; 512 = even
    // squared-Euclidean recall on the int8 screen (L2 = true, <= 128 queries): a pair is a suspect iff
    // I >= floor(A_q + B_q * min|x|^2 of the block) - 2, with thr_screen = A_q (float) and l2_b = B_q (screen_thr8_l2_kernel)
    const float* blk_nxmin;   // [blocks] smallest |x|^2 of every physical 32-row block (pg_table::d_nxmin)
    const float* l2_b;        // [256]
    const float* nx_rows;     // L2 = 2 (per-row test: g = 2 s_x s_q I - |x|^2 >= C'_q, thr_screen = C'_q, l2_b = 2 s_x s_q): |x|^2 of every row
#ifdef PG_SCREEN_PROFILE
    unsigned long long* prof; // [waves][8]
#endif
};

// Hit records (query halves).  A 32-row x 32-query tile with a suspect in it has, as a rule, exactly one: one lane, one of
// its 16 accumulators.  Finding the register inside the scan kernel — 16 compare / add-with-carry pairs and a staging
// loop, issued for the whole wave on behalf of that one lane — cost about as many VALU slots as the reject test itself.
// Instead the lane parks its 16 accumulators as they are (64 B) with a tag (first row of the block; query column and
// lane half) in the wave's LDS staging area and moves on: five LDS writes under the hit lanes' exec mask.  Every
// kRecStage records the area is copied to the wave's region of a global record buffer (plain coalesced stores, no
// atomics, no wait: the stores add to vmcnt, which only makes the ring's counted waits conservative for a moment —
// loads return in order among themselves).  screen_decode_kernel then does the compares with one record per lane — all
// 64 lanes busy — and fills the per-query suspect lists that rescore_kernel reads.  (Stores straight from the hit path,
// without the LDS stage, were measured: 4.35 vs 3.1 ms per pass — with stores in flight in every second block the
// counted waits over-wait all the time and the ring runs one piece deep.)
constexpr uint32_t kRecBytes = 80;
constexpr uint32_t kRecOvfWord = 1 + kMaxQueries + 1;       // rs.overflow[kRecOvfWord]: the overflow was one of the hit-record areas (they can grow: pg_table::rec_scale)
// make SCAN_EXTRA=-DPG_SCREEN_PROFILE: per-phase cycle counts of the query-halves loop, printed by launch_screen (developer aid)
#ifdef PG_SCREEN_PROFILE
#define SP_MARK(i) { const uint64_t tn = __builtin_readcyclecounter(); sp[i] += tn - sp_t; sp_t = tn; }
#else
#define SP_MARK(i)
#endif

// NQB query blocks of 32; WAVES waves per workgroup.  <=128 queries: 8 waves (2 per SIMD), ring of 4
// pieces per wave; 256 queries: the 256 B-operand registers leave room for one wave per SIMD only, so
// 4 waves with a ring of 8 pieces each.
// SPLIT = 2: waves (2j, 2j+1) walk the SAME run of blocks, each against its own half of the queries
// (NQB blocks each).  The table is then requested twice within a few microseconds; the second request
// is served by L2 / Infinity Cache, so HBM traffic stays ≈ 1x while both waves keep their B operand in
// 128 registers (two waves per SIMD) — 256 queries per pass without the one-wave-per-SIMD penalty.
// VAR: developer ablations (PG_SCAN_VARIANTS builds only; 0 = product; 1 = no screen test; 2 = no MFMA and
// no test; 3 = no DMA; 4 = test but never take the hit path) — they time the components, results are wrong
// I8: the shadow and the queries are int8 and the bound is an exact int32 dot product on
// v_mfma_i32_32x32x32_i8 — same issue rate as the bf16 MFMA at twice the k per instruction, over half the
// bytes per row.
// QH: query halves — the wave serves NQB x QH query blocks, QH groups of NQB one after the other on the same table
// block, re-using the accumulators (int8, 256 queries: 2 x 4 blocks in 8 waves — two waves per SIMD, so one
// wave's test, hit path and DMA issue run under the other's MFMAs; all 128 B-operand registers in the AGPR half).
template <int DIM, int NQB, int WAVES, int SPLIT = 1, int VAR = 0, bool I8 = false, int QH = 1, int L2 = 0>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void screen_kernel(ScreenArgs a) {
    static_assert(!L2 || (I8 && QH == 1 && SPLIT == 1), "squared-Euclidean screen: the int8 kernels of <= 128 queries");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the kernel streams the table's shadow: a piece is 32 rows x one 128-B line per row (64 bf16 / 128 int8)
    constexpr int EB = I8 ? 1 : 2;               // bytes per shadow element
    constexpr int kCols16 = 128 / EB;
    static_assert(DIM % kCols16 == 0, "a shadow row is a whole number of 128-B lines");
    constexpr int PPB = DIM / kCols16;           // pieces per 32-row block (1 or 2)
    constexpr int NS = kScanLdsRing / (WAVES * kPieceBytes);        // ring slots per wave
    constexpr int ND = kPieceDmas;
    constexpr int KS = DIM * EB / 32;            // k-steps: 16 bf16 or 32 int8 (32 B of a row) each
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    using AccT = typename std::conditional<I8, i32x16, f32x16>::type;
    using ThrT = typename std::conditional<I8, int, float>::type;
    // (query halves: the last two B fragments sit in LDS — 2 KiB for the workgroup — because the allocator wants a
    // few registers of the AGPR half for itself and would otherwise bring two fragments back from scratch per block,
    // behind an s_waitcnt vmcnt(0) that also drains the DMA ring)
    constexpr int kBLds = QH > 1 ? 2048 : 0;
    constexpr int kStageBytesW = (kScreenLds - kScanLdsRing - kBLds) / WAVES;
    constexpr int NQT = NQB * QH;                // query blocks of this wave
    static_assert(QH == 1 || (PPB == 1 && SPLIT == 1), "query halves: one piece per block");
    constexpr int kCap = (kStageBytesW - NQT * 256 - 16) / 8;       // staged (row, query) pairs per wave
    constexpr int kScanWaves = WAVES;            // (shadows the exact kernel's constant in this scope)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw_raw = blockIdx.x * kScanWaves + wave;
    const uint32_t gw = gw_raw / SPLIT;                          // block-run owner (shared by a wave group)
    const uint32_t W = gridDim.x * kScanWaves / SPLIT;
    const int qb0 = (int)(gw_raw % SPLIT) * NQT;                 // first query block of this wave
    const int i32 = lane & 31;
    const int h = lane >> 5;

    // B operand: bfrag[c][ks] = Q[c*32 + (lane&31)][ks*16 + 8h .. +7] as bf16
    uint4 bfrag[NQT][KS];
    ThrT thr_s[NQT];
    float eu[NQT];                                    // bf16: eps_unit of this lane's query, per query block
    float la[NQT];                                    // L2: A_q
    bool active[NQT];
    char* const b_lds = smem + kScreenLds - kBLds;
    if constexpr (QH > 1) {
        if (wave == 0) {
            *reinterpret_cast<uint4*>(b_lds + lane * 16) = a.qb16[((qb0 + NQT - 1) * KS + KS - 2) * 64 + lane];
            *reinterpret_cast<uint4*>(b_lds + 1024 + lane * 16) = a.qb16[((qb0 + NQT - 1) * KS + KS - 1) * 64 + lane];
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < NQT; ++c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (QH > 1 && c == NQT - 1 && ks >= KS - 2) bfrag[c][ks] = make_uint4(0, 0, 0, 0);     // (in LDS; unused)
            else bfrag[c][ks] = a.qb16[((qb0 + c) * KS + ks) * 64 + lane];
        }
        active[c] = (uint32_t)((qb0 + c) * 32 + i32) < a.nq;
        if constexpr (I8) thr_s[c] = active[c] ? __float_as_int(a.thr_screen[(qb0 + c) * 32 + i32]) : 0x7fffffff;
        else thr_s[c] = active[c] ? a.thr_screen[(qb0 + c) * 32 + i32] : __builtin_inff();
        eu[c] = (!I8 && active[c]) ? a.eps_unit[(qb0 + c) * 32 + i32] : 0.0f;
        if constexpr (L2 != 0) {                      // A_q in la, B_q in eu (L2 = 2: C'_q and 2 s_x s_q; inactive columns: never a suspect)
            la[c] = active[c] ? a.thr_screen[(qb0 + c) * 32 + i32] : __builtin_inff();
            eu[c] = active[c] ? a.l2_b[(qb0 + c) * 32 + i32] : 0.0f;
        }
    }
#pragma unroll
    for (int c = 0; c < NQT; ++c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (QH > 1 && c == NQT - 1 && ks >= KS - 2) continue;
            if (NQT > 4)     // >= 128 B registers: park them in the AGPR half, where the MFMA reads them directly
                asm volatile("" : "+a"(bfrag[c][ks].x), "+a"(bfrag[c][ks].y), "+a"(bfrag[c][ks].z), "+a"(bfrag[c][ks].w));
            else
                asm volatile("" : "+v"(bfrag[c][ks].x), "+v"(bfrag[c][ks].y), "+v"(bfrag[c][ks].z), "+v"(bfrag[c][ks].w));
        }
        asm volatile("" : "+v"(thr_s[c]));
        if (!I8 || L2) asm volatile("" : "+v"(eu[c]));
        if (L2) asm volatile("" : "+v"(la[c]));
    }

    uint32_t voff[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) {
        const int S = n * 64 + lane;
        const int i = S >> 3, p = S & 7;
        voff[n] = (uint32_t)(i * DIM * EB + 16 * ((p + (i >> 1)) & 7));      // row i, rotated 16-B quad p
    }
    const uint32_t lds_wave_u = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + wave * (NS * kPieceBytes));
    char* const lds_ptr = smem + wave * (NS * kPieceBytes);
    const int rd_row = i32 * 128;
    const int rd_rot = (i32 >> 1) * 16;

    const uint32_t total = a.rb_end - a.rb_begin;
    uint32_t first, nblk;
    if (WAVES == 8 && SPLIT == 1 && a.early_share != 512) {
        // Two waves per SIMD (w and w + 4), and the SIMD's arbiter favours the older one: per-phase cycle counts of the
        // 256-query kernel had waves 0-3 finish an equal share in 78 % of the time of waves 4-7, which then ran the last fifth
        // of the launch alone — one wave per SIMD, nothing to overlap with.  So the pair's run of blocks is split unevenly
        // (early_share / 1024 of it to the early wave) and both finish together.
        const uint32_t P = gridDim.x * 4;
        const uint32_t pb = (total + P - 1) / P;
        const uint32_t pbase = (blockIdx.x * 4 + (wave & 3)) * pb;
        const uint32_t pend = pbase + pb < total ? pbase + pb : total;
        uint32_t cut = pbase + (uint32_t)(((uint64_t)pb * a.early_share + 512) >> 10);
        if (cut > pend) cut = pend;
        first = wave < 4 ? pbase : cut;
        const uint32_t last = wave < 4 ? cut : pend;
        nblk = first < last ? last - first : 0;
    } else {
        const uint32_t bpw = (total + W - 1) / W;
        first = gw * bpw;
        nblk = first < total ? (total - first < bpw ? total - first : bpw) : 0;
    }
    if (nblk == 0) {
        if (QH > 1 && lane == 0) a.rec_cnt[gw_raw] = 0;
        return;
    }

    // Physical table block of each logical block this wave walks, kept D blocks ahead of the one being
    // scored (the DMA ring runs NS-1 pieces ahead).  Incremental: slot += mul (mod n_slots) per block
    // for a pilot sample, +1 for a full pass — no division in the loop.  Blocks past the end of the
    // run repeat the last one (tail pieces re-read it; they are never scored).
    // (Tried for the query halves: the next piece sent into the slot just READ — its four fragments are in registers at the top
    // of the iteration — i.e. the whole ring, NS pieces = 128 KiB per CU, in flight instead of NS - 1: 311 vs 318 M items/s.
    // More requests in flight than the memory pipeline queues only stall the wave at the DMA issue.)
    constexpr int D = (PPB - 1 + NS - 1) / PPB;
    const bool strided = a.stride != 1;
    const uint32_t L0 = a.rb_begin + first;
    uint32_t walk_slot = strided ? (uint32_t)(((uint64_t)L0 * a.perm_mul) % a.perm_mod) : L0;
    uint32_t walk_n = 0;                          // logical blocks handed out so far
    auto next_phys = [&]() -> uint32_t {
        const uint32_t ph = strided ? slot_block(walk_slot, a.stride) : walk_slot;
        if (walk_n + 1 < nblk) {
            ++walk_n;
            if (strided) {
                walk_slot += a.perm_mul;
                if (walk_slot >= a.perm_mod) walk_slot -= a.perm_mod;
            } else {
                ++walk_slot;
            }
        }
        return __builtin_amdgcn_readfirstlane(ph);
    };
    uint32_t phys[D + 1];
#pragma unroll
    for (int j = 0; j <= D; ++j) phys[j] = next_phys();
    // wave-uniform source base and LDS destination of piece `pc_abs` pieces after the start of the
    // block being scored (compile-time offset), global piece counter t for the ring slot
    auto piece_addr = [&](int rel_piece, uint32_t t, const char*& ub, uint32_t& dst) {
        const uint32_t ph = phys[rel_piece / PPB];
        const char* base = (const char*)a.tab16 + (uint64_t)ph * (uint64_t)(kPieceRows * DIM * EB) + (rel_piece % PPB) * 128;
        const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)(uintptr_t)base >> 32));
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
        ub = (const char*)(((uint64_t)hi << 32) | lo);
        dst = lds_wave_u + __builtin_amdgcn_readfirstlane(t % NS) * kPieceBytes;
    };

    // staging: (row, query) pairs that passed the screen ("suspects").  They are parked in a
    // wave-private LDS list and flushed in bulk into per-query global suspect lists; the exact
    // re-scoring runs afterwards in rescore_kernel, with the whole chip hiding the gather latency
    // (done here it cost one dependent HBM round trip chain per 64 suspects per wave).
    uint32_t* const st_row = reinterpret_cast<uint32_t*>(smem + kScanLdsRing + wave * kStageBytesW);
    uint32_t* const st_q = st_row + kCap;
    uint32_t* const st_cnt = st_q + kCap;                 // [NQT*32]
    uint32_t* const st_base = st_cnt + NQT * 32;          // [NQT*32]
    uint32_t st_n = 0;
    auto flush = [&]() {
        // per-query reservation (one returning atomic per query present), then scatter the row ids
#pragma unroll
        for (int part = 0; part < (NQT + 1) / 2; ++part)
            if (lane + 64 * part < NQT * 32) st_cnt[lane + 64 * part] = 0;
        for (uint32_t e0 = 0; e0 < st_n; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < st_n) atomicAdd(&st_cnt[st_q[e]], 1u);
        }
#pragma unroll
        for (int part = 0; part < (NQT + 1) / 2; ++part) {
            const int q = lane + 64 * part;
            if (q < NQT * 32) {
                const uint32_t c = st_cnt[q];
                st_base[q] = c ? atomicAdd(&a.susp_cnt[qb0 * 32 + q], c) : 0u;
                st_cnt[q] = 0;
            }
        }
        for (uint32_t e0 = 0; e0 < st_n; e0 += 64) {
            const uint32_t e = e0 + lane;
            if (e < st_n) {
                const uint32_t q = st_q[e];
                const uint32_t pos = st_base[q] + atomicAdd(&st_cnt[q], 1u);
                if (pos < a.cap) a.susp[(uint64_t)(qb0 * 32 + q) * a.cap + pos] = st_row[e];
                else *a.overflow = 1u;
            }
        }
        st_n = 0;
        __builtin_amdgcn_s_waitcnt(0x0F70);
    };

#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        const char* src;
        uint32_t dst;
        piece_addr(t, t, src, dst);
#pragma unroll
        for (int n = 0; n < ND; ++n) dma_one(src, dst + n * 1024, voff[n], 0.0f);
    }

    if constexpr (QH > 1) {
        // hit records: staged in LDS (the area the other variants stage (row, query) pairs in), flushed to this wave's region
        constexpr uint32_t kRecStage = (uint32_t)kStageBytesW / kRecBytes;
        char* const rec_lds = smem + kScanLdsRing + wave * kStageBytesW;
        char* const rec_glb = a.rec + (size_t)gw_raw * a.rec_cap * kRecBytes;
        uint32_t rec_st = 0, rec_total = 0;
        auto rec_flush = [&]() {
            if (rec_total + rec_st > a.rec_cap) {
                // region full (the table's best rows sit together): the staged records go to the shared pool — one returning
                // atomic, which also drains the ring; rare by construction.  A full pool fails the plan (the caller falls back).
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(&a.rec_cnt[a.rec_waves], rec_st);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g + rec_st > a.rec_pool_cap) {
                    if (lane == 0) { *a.overflow = 1u; a.overflow[kRecOvfWord] = 1u; }
                } else {
                    char* const dst = a.rec_pool + (size_t)g * kRecBytes;
                    for (uint32_t o = lane * 16; o < rec_st * kRecBytes; o += 1024)
                        *reinterpret_cast<i32x4*>(dst + o) = *reinterpret_cast<const i32x4*>(rec_lds + o);
                }
            } else {
                char* const dst = rec_glb + (size_t)rec_total * kRecBytes;
                for (uint32_t o = lane * 16; o < rec_st * kRecBytes; o += 1024)
                    *reinterpret_cast<i32x4*>(dst + o) = *reinterpret_cast<const i32x4*>(rec_lds + o);
                rec_total += rec_st;
            }
            rec_st = 0;
        };
#ifdef PG_SCREEN_PROFILE
        uint64_t sp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sp_t = __builtin_readcyclecounter();
#endif
        for (uint32_t b = 0; b < nblk; ++b) {
            if (VAR != 3 && VAR != 5) wait_vmcnt<ND * (NS - 2)>();
            SP_MARK(0)
            const char* nb_src;
            uint32_t nb_dst;
            piece_addr(NS - 1, b + NS - 1, nb_src, nb_dst);
            const char* slot = lds_ptr + (b % NS) * kPieceBytes + rd_row;
            f32x4 q4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) q4[j] = *reinterpret_cast<const f32x4*>(slot + (((2 * j + h) * 16 - rd_rot) & 112));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < ND; ++n)
                if (VAR != 3 && VAR != 5) dma_one(nb_src, nb_dst + n * 1024, voff[n], q4[n].x);
            SP_MARK(1)
            const uint32_t cur_phys = phys[0];
#pragma unroll
            for (int j = 0; j < D; ++j) phys[j] = phys[j + 1];
            phys[D] = next_phys();
            const uint32_t row0 = cur_phys * kPieceRows;
#pragma unroll
            for (int half = 0; half < QH; ++half) {
                AccT acc[NQB];
#pragma unroll
                for (int c = 0; c < NQB; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[c][r] = 0;
#pragma unroll
                for (int ksl = 0; ksl < 4; ++ksl)
#pragma unroll
                    for (int c = 0; c < NQB; ++c) {
                        if (VAR == 2) {                      // (ablation: stream only)
                            asm volatile("" :: "v"(q4[ksl]));
                            continue;
                        }
                        uint4 bq;
                        if (half * NQB + c == NQT - 1 && ksl >= KS - 2) bq = *reinterpret_cast<const uint4*>(b_lds + (ksl - (KS - 2)) * 1024 + lane * 16);
                        else bq = bfrag[half * NQB + c][ksl];
                        acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, q4[ksl]),
                                                                       __builtin_bit_cast(i32x4, bq), acc[c], 0, 0, 0);
                    }
                if (VAR == 1 || VAR == 2 || VAR == 5) {      // (ablation: no screen test)
#pragma unroll
                    for (int c = 0; c < NQB; ++c) asm volatile("" :: "v"(acc[c][0]), "v"(acc[c][15]));
                    continue;
                }
                uint64_t cmask[NQB];
                bool hit[NQB];
                uint64_t any_mask = 0;
#pragma unroll
                for (int c = 0; c < NQB; ++c) {
                    int m = acc[c][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = acc[c][r] > m ? acc[c][r] : m;
                    hit[c] = m >= thr_s[half * NQB + c];
                    cmask[c] = __builtin_amdgcn_ballot_w64(hit[c]);
                    any_mask |= cmask[c];
                }
                if (VAR == 4) {                              // (ablation: test, never the hit path)
                    asm volatile("" :: "s"(any_mask));
                    any_mask = 0;
                }
                SP_MARK(2 + 2 * half)
                if (any_mask != 0) {
#pragma unroll
                    for (int c = 0; c < NQB; ++c) {
                        if (cmask[c] == 0) continue;
                        uint64_t pend = cmask[c];
                        bool mine = hit[c];
                        for (;;) {
                            const uint32_t n = (uint32_t)__popcll(pend);
                            if (rec_st + n > kRecStage && rec_st != 0) rec_flush();
                            const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(pend >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)pend, 0u));
                            const bool take = mine && pre < kRecStage;      // (more than kRecStage hit lanes: in rounds)
                            if (take) {
                                char* const r = rec_lds + (rec_st + pre) * kRecBytes;
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) {
                                    const i32x4 v = {acc[c][4 * jj], acc[c][4 * jj + 1], acc[c][4 * jj + 2], acc[c][4 * jj + 3]};
                                    *reinterpret_cast<i32x4*>(r + 16 * jj) = v;
                                }
                                *reinterpret_cast<uint2*>(r + 64) = make_uint2(row0, (uint32_t)((qb0 + half * NQB + c) * 32 + i32) | ((uint32_t)h << 8));
                            }
                            rec_st += n < kRecStage ? n : kRecStage;
                            if (n <= kRecStage) break;
                            mine = mine && !take;
                            pend = __builtin_amdgcn_ballot_w64(mine);
                        }
                    }
                }
                SP_MARK(3 + 2 * half)
            }
        }
        rec_flush();
        if (lane == 0) a.rec_cnt[gw_raw] = rec_total;
#ifdef PG_SCREEN_PROFILE
        if (lane == 0 && a.prof)
            for (int i = 0; i < 8; ++i) a.prof[gw_raw * 8 + i] = sp[i];
#endif
    } else {
    for (uint32_t b = 0; b < nblk; ++b) {
        AccT acc[NQB];
#pragma unroll
        for (int c = 0; c < NQB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0;
#pragma unroll
        for (int pc = 0; pc < PPB; ++pc) {
            const uint32_t t = b * PPB + pc;
            if (VAR != 3) wait_vmcnt<ND * (NS - 2)>();
            const char* nb_src;
            uint32_t nb_dst;
            piece_addr(pc + NS - 1, t + NS - 1, nb_src, nb_dst);
            const char* slot = lds_ptr + (t % NS) * kPieceBytes + rd_row;
            // A fragment of k-step ksl (32 B of the row = quads 2ksl, 2ksl+1: 16 bf16 or 32 int8 columns): this
            // lane's half is quad 2ksl + h — one ds_read_b128 is one MFMA operand, no conversion
            f32x4 q4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int quad = 2 * j + h;
                q4[j] = *reinterpret_cast<const f32x4*>(slot + ((quad * 16 - rd_rot) & 112));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < ND; ++n)
                if (VAR != 3) dma_one(nb_src, nb_dst + n * 1024, voff[n], q4[n].x);
#pragma unroll
            for (int ksl = 0; ksl < 4; ++ksl) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, q4[ksl]);
                const int ks = pc * 4 + ksl;
                if (VAR == 2) {
                    asm volatile("" :: "v"(af));
                    continue;
                }
#pragma unroll
                for (int c = 0; c < NQB; ++c) {
                    if constexpr (I8)
                        acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, q4[ksl]),
                                                                       __builtin_bit_cast(i32x4, bfrag[c][ks]), acc[c], 0, 0, 0);
                    else
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bfrag[c][ks]), acc[c], 0, 0, 0);
                }
            }
        }
        const uint32_t cur_phys = phys[0];
#pragma unroll
        for (int j = 0; j < D; ++j) phys[j] = phys[j + 1];
        phys[D] = next_phys();
        // ---- screen test.  Each lane packs its hits into a bit set (bit c*16+r ↔ accumulator c,
        // register r); the slow path below only needs (row, query) — both follow from the bit index —
        // so the accumulators are never indexed dynamically.
        if (VAR == 1 || VAR == 2) {
#pragma unroll
            for (int c = 0; c < NQB; ++c) asm volatile("" :: "v"(acc[c][0]), "v"(acc[c][15]));
            continue;
        }
        // Block reject: the largest of a lane's 16 bounds per query block against the screen threshold
        // (7 max3 + 1 max + 1 compare per query block).  fmaxf drops a NaN operand, which is safe: the
        // accumulators of a finite table and a finite, moderate query are finite, and every other query
        // has thr_screen = -inf (screen_thr_kernel) so that everything passes.  Inactive query columns
        // carry thr_s = +inf (and eps_unit 0).
        // bf16: this block's cutoffs, cut = thr - eps_unit x (largest row norm of the block), rounded down (the margin
        // term is kept finite so that +-inf thresholds stay what they are)
        ThrT cut[NQB];
        float nxr[16];                                   // L2 = 2: |x|^2 of this lane's 16 rows of the block
        if constexpr (L2 == 2) {
            const float* const nxp = a.nx_rows + (size_t)cur_phys * kPieceRows;      // wave-uniform: scalar loads
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const float nxa = nxp[ro], nxb = nxp[ro + 4];
                nxr[r] = h ? nxb : nxa;
            }
#pragma unroll
            for (int c = 0; c < NQB; ++c) cut[c] = 0;
        } else if constexpr (L2 == 1) {
            // the block's integer cutoffs: floor(A_q + B_q * (smallest |x|^2 of the block)) - 2 (rounded down twice over;
            // -inf / NaN / below the int range: everything is a suspect)
            const float nxm = a.blk_nxmin[cur_phys];
#pragma unroll
            for (int c = 0; c < NQB; ++c) {
                const float v = __fmaf_rn(eu[c], nxm, la[c]);
                // (-inf or NaN: everything is a suspect; +inf — an inactive query column — or beyond the int range: nothing is)
                cut[c] = !(v > -2.0e9f) ? (int)0x80000000
                                        : (v >= 2.0e9f ? 0x7fffffff : __float2int_rd(v - fabsf(v) * 2.4e-7f - 2.0f));
            }
        } else if constexpr (I8) {
#pragma unroll
            for (int c = 0; c < NQB; ++c) cut[c] = thr_s[c];
        } else {
            const float nb = sqrtf(a.blk_norm2[cur_phys]) * 1.0002f;
#pragma unroll
            for (int c = 0; c < NQB; ++c) {
                const float v = __fmaf_rn(-eu[c], nb, thr_s[c]);
                cut[c] = v - __fmaf_rn(fminf(fabsf(v), 3.0e38f), 1.2e-7f, 1e-37f);
            }
        }
        uint64_t cmask[NQB];
        uint64_t any_mask = 0;
#pragma unroll
        for (int c = 0; c < NQB; ++c) {
            if constexpr (L2 == 2) {
                // per-ROW test: 2 s_x s_q I - |x|^2 against C'_q, in fp32 (|I| < 2^24 converts exactly; the roundings are in C'_q)
                float m = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, __fmaf_rn((float)acc[c][r], eu[c], -nxr[r]));
                cmask[c] = __builtin_amdgcn_ballot_w64(!(m < la[c]));
            } else if constexpr (I8) {
                int m = acc[c][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = acc[c][r] > m ? acc[c][r] : m;
                cmask[c] = __builtin_amdgcn_ballot_w64(m >= cut[c]);
            } else {
                float m = acc[c][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[c][r]);
                cmask[c] = __builtin_amdgcn_ballot_w64(!(m < cut[c]));
            }
            any_mask |= cmask[c];
        }
        if (VAR == 4) any_mask = 0;
        if (any_mask != 0) {
            // the rare block with a hit: only the query blocks that have one are looked at again
            const uint32_t row0 = cur_phys * kPieceRows;
#pragma unroll
            for (int c = 0; c < NQB; ++c) {
                if (cmask[c] == 0) continue;
                // per-lane bit set of this query block: bit 15 - r ↔ register r (the compare sets VCC,
                // add-with-carry shifts it in: m = 2m + pass).  The accumulators were all read by the max
                // chains above, so the MFMA → VALU hazard the compiler cannot see inside asm is covered.
                uint32_t m16 = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (L2 == 2) {
                        const float g = __fmaf_rn((float)acc[c][r], eu[c], -nxr[r]);
                        asm volatile("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                     : "+v"(m16) : "v"(g), "v"(la[c]) : "vcc");
                    } else if constexpr (I8)
                        asm volatile("v_cmp_ge_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                     : "+v"(m16) : "v"(acc[c][r]), "v"(cut[c]) : "vcc");
                    else
                        asm volatile("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                                     : "+v"(m16) : "v"(acc[c][r]), "v"(cut[c]) : "vcc");
                }
                if (row0 + kPieceRows > a.row_end) {        // last block of a ragged table: drop rows past the end
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (row0 + (r & 3) + 8 * (r >> 2) + 4 * h >= a.row_end) m16 &= ~(1u << (15 - r));
                }
                // stage the hits, one per lane per round (ballot + prefix count: no LDS atomics, no waits)
                for (;;) {
                    const bool p = m16 != 0;
                    const uint64_t bm = __builtin_amdgcn_ballot_w64(p);
                    if (bm == 0) break;
                    const uint32_t n = __popcll(bm);
                    if (st_n + n > (uint32_t)kCap) flush();
                    if (p) {
                        const int r = 15 - __builtin_ctz(m16);
                        const uint32_t pos = st_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        st_row[pos] = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        st_q[pos] = (uint32_t)(c * 32 + i32);
                        m16 &= m16 - 1;
                    }
                    st_n += n;
                }
            }
        }
    }
    }
    if constexpr (QH == 1) flush();
    wait_vmcnt<0>();
}

// hit records → per-query suspect lists (see kRecBytes).  One workgroup per scan workgroup (its eight wave regions), one
// record per thread and round.  Two passes over the records: count per query in LDS, ONE global atomic per query and
// workgroup to reserve the range (one per suspect — 2.3 M returning atomics on 256 counters — made this kernel take 1.08 ms),
// then place the rows.
__device__ __forceinline__ uint32_t rec_mask(const char* r, const float* __restrict__ thr_screen, uint32_t row_end, uint32_t& q,
                                             uint32_t& row0h) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const uint2 tag = *reinterpret_cast<const uint2*>(r + 64);
    q = tag.y & 255u;
    const uint32_t h = (tag.y >> 8) & 1u;
    row0h = tag.x + 4 * h;
    const int t = __float_as_int(thr_screen[q]);
    uint32_t m = 0;                                        // bit rr: accumulator register rr of the lane is a suspect
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const i32x4 v = *reinterpret_cast<const i32x4*>(r + 16 * jj);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rr = 4 * jj + e;
            if (v[e] >= t && row0h + (rr & 3) + 8 * (rr >> 2) < row_end) m |= 1u << rr;
        }
    }
    return m;
}
constexpr uint32_t kRecPoolSlice = 8192;       // records of the spill pool per decode workgroup
__global__ __launch_bounds__(1024) void screen_decode_kernel(const char* __restrict__ rec, const uint32_t* __restrict__ rec_cnt,
                                                             uint32_t rec_cap, uint32_t rec_waves, const char* __restrict__ rec_pool,
                                                             uint32_t rec_pool_cap, const float* __restrict__ thr_screen, uint32_t row_end,
                                                             uint32_t* __restrict__ susp_cnt, uint32_t* __restrict__ susp,
                                                             uint32_t cap, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t cntq[kMaxQueries], baseq[kMaxQueries], nreg[8];
    // pass 1 leaves (first row + lane half, query | mask << 16) of every record with a suspect here, pass 2 reads them back
    // (records beyond the buffer — a workgroup of a skewed table — are read from global memory again)
    constexpr uint32_t kKeep = 12288;
    __shared__ uint2 keep[kKeep];
    // workgroups [0, rec_waves / 8): the eight wave regions of one scan workgroup; the ones behind: a slice of the spill pool
    const bool pool = blockIdx.x >= rec_waves / 8;
    if (threadIdx.x < kMaxQueries) cntq[threadIdx.x] = 0;
    if (threadIdx.x < 8) {
        uint32_t n;
        if (pool) {
            const uint32_t filled = rec_cnt[rec_waves] < rec_pool_cap ? rec_cnt[rec_waves] : rec_pool_cap;
            const uint32_t begin = (blockIdx.x - rec_waves / 8) * kRecPoolSlice;
            n = threadIdx.x == 0 && begin < filled ? (filled - begin < kRecPoolSlice ? filled - begin : kRecPoolSlice) : 0u;
        } else {
            const uint32_t n_raw = rec_cnt[blockIdx.x * 8 + threadIdx.x];
            n = n_raw < rec_cap ? n_raw : rec_cap;
        }
        nreg[threadIdx.x] = n;
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t off = 0;
        for (uint32_t w = 0; w < 8; ++w) {
            const uint32_t n = nreg[w];
            const char* const base = pool ? rec_pool + (size_t)(blockIdx.x - rec_waves / 8) * kRecPoolSlice * kRecBytes
                                          : rec + (size_t)(blockIdx.x * 8 + w) * rec_cap * kRecBytes;
            for (uint32_t e = threadIdx.x; e < n; e += 1024) {
                uint32_t q, row0h, m;
                const uint32_t slot = off + e;
                if (pass == 1 && slot < kKeep) {
                    const uint2 kp = keep[slot];
                    row0h = kp.x;
                    q = kp.y & 0xffffu;
                    m = kp.y >> 16;
                } else {
                    m = rec_mask(base + (size_t)e * kRecBytes, thr_screen, row_end, q, row0h);
                    if (pass == 0 && slot < kKeep) keep[slot] = make_uint2(row0h, q | (m << 16));
                }
                if (m == 0) continue;
                const uint32_t nh = __popc(m);
                if (pass == 0) {
                    atomicAdd(&cntq[q], nh);
                } else {
                    uint32_t pos = baseq[q] + atomicAdd(&cntq[q], nh);
                    if (pos + nh > cap) { *overflow = 1u; continue; }
                    while (m) {
                        const uint32_t rr = __builtin_ctz(m);
                        m &= m - 1;
                        susp[(uint64_t)q * cap + pos++] = row0h + (rr & 3) + 8 * (rr >> 2);
                    }
                }
            }
            off += n;
        }
        __syncthreads();
        if (pass == 0 && threadIdx.x < kMaxQueries) {
            const uint32_t c = cntq[threadIdx.x];
            baseq[threadIdx.x] = c ? atomicAdd(&susp_cnt[threadIdx.x], c) : 0u;
            cntq[threadIdx.x] = 0;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// rescore: exact scores of the suspects a screened launch parked in susp[q][0..susp_cnt[q]) — the
// specification's k-ascending fmaf chain over the fp32 table row and the fp32 query — keeping those
// with !(s < thr[q]) as candidate keys in cand[q].  One block = 256 suspects of ONE query (grid.y = query,
// grid.x strides over the list): the query sits in LDS (broadcast reads), each wave gathers its 64 rows
// with coalesced 256 B segments into a padded LDS tile (16 independent loads per lane in flight), then
// lane s walks row s.  One returning atomic per block reserves the output range.
// rescore_kernel's two steps on one unit (a 256-B half of 64 rows): the gather of the wave's 64 rows (16
// independent 16-B loads per lane, four rows per instruction) and the walk (LDS tile, then lane s runs the
// specification's k-ascending fmaf chain over row s)
template <int DIM>
__device__ __forceinline__ void rescore_gather(f32x4 (&buf)[16], const float* __restrict__ tab, uint32_t row, int ph, int lane) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t r_i = (uint32_t)__shfl((int)row, 4 * i + (lane >> 4), 64);
        buf[i] = reinterpret_cast<const f32x4*>(tab + (size_t)r_i * DIM + ph * 64)[lane & 15];
    }
}
template <int ROWB>
__device__ __forceinline__ void rescore_walk(const f32x4 (&buf)[16], char* my, const float* qs, int ph, int lane, float& s) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
        *reinterpret_cast<f32x4*>(my + (4 * i + (lane >> 4)) * ROWB + (lane & 15) * 16) = buf[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float4 x = *reinterpret_cast<const float4*>(my + lane * ROWB + k * 16);
        const float4 y = *reinterpret_cast<const float4*>(&qs[ph * 64 + 4 * k]);
        s = __fmaf_rn(x.x, y.x, s);
        s = __fmaf_rn(x.y, y.y, s);
        s = __fmaf_rn(x.z, y.z, s);
        s = __fmaf_rn(x.w, y.w, s);
        // (keeps hipcc from hoisting all 32 LDS reads above the chain: with two gathers in flight that pushed the
        // kernel past 256 registers and one gather buffer into scratch)
        if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_wave_barrier();
}

// (L2: the exact inner product is turned into -d = fmaf(2, ip, -(|x|^2 + |q|^2)) before the threshold test and the key)
template <int DIM, bool L2 = false>
__global__ __launch_bounds__(256) void rescore_kernel(const float* __restrict__ tab, const float* __restrict__ qpad,
                                                      const float* __restrict__ thr,
                                                      const uint32_t* __restrict__ susp,
                                                      const uint32_t* __restrict__ susp_cnt, uint32_t cap,
                                                      uint32_t* __restrict__ cnt, uint64_t* __restrict__ cand,
                                                      uint32_t* __restrict__ overflow, uint32_t scap, uint32_t n_rows,
                                                      const float* __restrict__ nx = nullptr, const float* __restrict__ nqv = nullptr,
                                                      RowFilter filter = RowFilter()) {
    // (scap: capacity and stride of the suspect lists; cap: of the candidate lists)
    constexpr int kRowB = 64 * 4 + 16;                 // 64 columns per phase, padded: conflict-free b128 column walks
    __shared__ __attribute__((aligned(16))) char tile[4][64 * kRowB];
    __shared__ __attribute__((aligned(16))) float qs[DIM];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t base_s;
    const uint32_t q = blockIdx.y;
    const uint32_t n_raw = susp_cnt[q];
    const uint32_t n = n_raw < scap ? n_raw : scap;
    if (blockIdx.x * 256u >= n) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < DIM) qs[threadIdx.x] = qpad[(size_t)q * DIM + threadIdx.x];
    const float thr_q = thr[q];
    __syncthreads();
    char* const my = tile[w];
    // The gathers run one unit (a 256-B half of 64 rows) ahead of the arithmetic: while a wave walks one half-row
    // tile, the loads of the next half — or of the next chunk's first half — are in flight (one exposed HBM round
    // trip per chunk and phase before).
    // (a list that overflowed has holes — the decode kernel reserves a run of slots and writes none of it when the run crosses
    //  the capacity — whose stale contents may be rows of an earlier, larger table: the plan is discarded, but the gather must
    //  stay inside this table.  Found as a memory fault by scripts/soak_adversarial.py: zero queries, every row a tie.)
    auto row_of = [&](uint32_t t0) -> uint32_t {
        const uint32_t e = t0 + threadIdx.x;
        const uint32_t r = e < n ? susp[(uint64_t)q * scap + e] : 0u;
        return r < n_rows ? r : 0u;
    };
    constexpr int NPH = DIM / 64;
    const uint32_t step = gridDim.x * 256u;
    uint32_t t0 = blockIdx.x * 256u;
    uint32_t row = row_of(t0);
    f32x4 va[16], vb[16];              // (ext vectors: HIP's float4 struct made these loop-carried buffers memcpy'd stack objects)
    rescore_gather<DIM>(va, tab, row, 0, lane);
    for (; t0 < n; t0 += step) {
        const bool valid = t0 + threadIdx.x < n;
        const uint32_t row_next = t0 + step < n ? row_of(t0 + step) : 0u;
        float s = 0.0f;
        // (phases written out: with `v[ph & 1]` inside a loop the buffers stayed in scratch at DIM = 128; past the
        // last chunk row_next is 0 — a harmless gather of row 0 instead of a conditional one)
        if constexpr (NPH == 2) {
            rescore_gather<DIM>(vb, tab, row, 1, lane);
            rescore_walk<kRowB>(va, my, qs, 0, lane, s);
            rescore_gather<DIM>(va, tab, row_next, 0, lane);
            rescore_walk<kRowB>(vb, my, qs, 1, lane, s);
        } else {
            rescore_gather<DIM>(vb, tab, row_next, 0, lane);
            rescore_walk<kRowB>(va, my, qs, 0, lane, s);
        }
        if (NPH & 1) {                                   // odd phase count: the prefetched unit sits in v[1], the loop reads v[0]
#pragma unroll
            for (int i = 0; i < 16; ++i) va[i] = vb[i];
        }
        if constexpr (L2) s = __fmaf_rn(2.0f, s, -(nx[row] + nqv[q]));
        const bool keep = valid && !(s < thr_q) && row_filter_pass(filter, row);
        const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            base_s = tot ? atomicAdd(&cnt[q], tot) : 0u;
        }
        __syncthreads();
        uint32_t pos = base_s + before;
        for (int j = 0; j < w; ++j) pos += wsum[j];
        if (keep) {
            if (pos < cap) cand[(uint64_t)q * cap + pos] = topk_key(s, row);
            else *overflow = 1u;
        }
        __syncthreads();
        row = row_next;
    }
    if (n_raw > scap && threadIdx.x == 0) *overflow = 1u;
}

// per call: bf16 B fragments of the (zero-padded) queries and eps_unit_q = kScreenEps * ||q|| (the screen multiplies it
// by the largest row norm of each 32-row block: the bf16 bound is relative to the row, so one huge row does not
// loosen it for the rest of the table)
__global__ void screen_prep_kernel(const float* __restrict__ qpad, uint32_t dim,
                                   uint4* __restrict__ qb16, float* __restrict__ eps) {
    const uint32_t KS = dim / 16;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (uint32_t)kScreenMaxNQB * KS * 64) {
        const uint32_t lane = i & 63, ks = (i >> 6) % KS, c = (i >> 6) / KS;
        const float* q = qpad + (size_t)(c * 32 + (lane & 31)) * dim + ks * 16 + 8 * (lane >> 5);
        uint32_t w[4];
        for (int e = 0; e < 4; ++e) {
            auto rne = [](float x) -> uint32_t {
                uint32_t b = __float_as_uint(x);
                if ((b & 0x7FFFFFFFu) > 0x7F800000u) return (b >> 16) | 0x40u;
                b += 0x7FFFu + ((b >> 16) & 1u);
                return b >> 16;
            };
            w[e] = rne(q[2 * e]) | (rne(q[2 * e + 1]) << 16);
        }
        qb16[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (i < (uint32_t)kMaxQueries) {
        double ss = 0.0;
        for (uint32_t k = 0; k < dim; ++k) {
            const double v = (double)qpad[(size_t)i * dim + k];
            ss += v * v;
        }
        // inflated so that it is an upper bound in fp32
        eps[i] = (float)(sqrt(ss) * (double)kScreenEps * 1.0001) + 1e-30f;
    }
}

// bf16 screen: thr_screen = thr (the margin is applied per block in the kernel), or -inf
__global__ void screen_thr_kernel(const float* __restrict__ thr, const float* __restrict__ eps, float max_norm,
                                  float* __restrict__ thr_screen) {
    const uint32_t q = threadIdx.x;
    if (q >= (uint32_t)kMaxQueries) return;
    const float t = thr[q], e = eps[q];
    // the kernel subtracts eps_unit x block norm itself; non-finite queries, or magnitudes whose bf16 partial sums
    // could overflow — |partial sum| <= ||x|| ||q|| = 125 eps_unit ||x||, and an inf - inf accumulator would be a NaN
    // that the block-reject max chain drops silently — : everything passes the screen and the exact re-scoring decides
    float v = t;
    if (!(e == e) || e > 1e28f || !(t == t) || !(e * max_norm < 1e35f)) v = -__builtin_inff();
    thr_screen[q] = v;
}

// ---- int8 screen (dim 128).  x^ = s_x X, q^ = s_q Q with X, Q in [-127, 127]; the MFMA gives the integer dot
// product I = sum X_i Q_i exactly.  For the true score s = sum x_i q_i (real arithmetic):
//     s - s_x s_q I = sum (x_i - x^_i) q_i + sum x^_i (q_i - q^_i)
//     |s - s_x s_q I| <= ||x - x^|| ||q|| + ||x^|| ||q - q^||  <=  R ||q|| + (N + R) ||q - q^||
// with R = max row residual (pg_table::resid8, measured when the shadow is built), N = max row norm.  eps_q is
// that bound plus 1e-5 N ||q|| for the rounding of the specification's fp32 fmaf chain (<= 128 * 2^-24 N ||q||).
// A row can reach thr only if  I >= (thr - eps_q) / (s_x s_q): the screen compares integers.
// per call: int8 B fragments of the (zero-padded) queries, eps_q and the per-query scale s_q = max|q| / 127
// One workgroup per query block of 32 in use (the other slots of the 256 are neither read by the scan nor prepared:
// a single request used to spend 36 us preparing 255 zero queries, a full batch 37 us in one 1024-thread workgroup).
__global__ __launch_bounds__(128) void screen_prep8_kernel(const float* __restrict__ qpad, uint32_t dim,
                                                            float max_norm, float resid, uint4* __restrict__ qb8,
                                                            float* __restrict__ eps, float* __restrict__ qscale) {
    __shared__ float sq[32];
    const uint32_t tid = threadIdx.x, c = blockIdx.x;
    {
        // four threads per query, a quarter of the columns each
        const uint32_t ql = tid >> 2, qi = c * 32 + ql, part = tid & 3, per = dim / 4;
        const float* q = qpad + (size_t)qi * dim + part * per;
        float mx = 0.0f;
        bool bad = false;
        for (uint32_t k = 0; k < per; ++k) {
            const float v = fabsf(q[k]);
            if (!(v <= 3.0e38f)) bad = true;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        bad = (__shfl_xor((int)bad, 1, 64) | (int)bad) != 0;
        bad = (__shfl_xor((int)bad, 2, 64) | (int)bad) != 0;
        const float sc = fmaxf(mx / 127.0f, 1e-30f);
        double ss = 0.0, dd = 0.0;
        for (uint32_t k = 0; k < per; ++k) {
            const double v = (double)q[k];
            int Q = __float2int_rn(q[k] / sc);
            Q = Q > 127 ? 127 : (Q < -127 ? -127 : Q);
            const double d = v - (double)sc * (double)Q;
            ss += v * v;
            dd += d * d;
        }
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        dd += __shfl_xor(dd, 1, 64);
        dd += __shfl_xor(dd, 2, 64);
        if (part == 0) {
            // (sums of non-negative terms: any summation order is an upper bound once inflated below)
            const double nq = sqrt(ss), dq = sqrt(dd);
            const double e = ((double)resid * nq + ((double)max_norm + (double)resid) * dq) * 1.0001 +
                             1e-5 * (double)max_norm * nq + 1e-30;
            eps[qi] = bad ? __builtin_nanf("") : (float)(e * 1.000001);      // upper bound in fp32
            qscale[qi] = sc;
            sq[ql] = sc;
        }
    }
    __syncthreads();
    const uint32_t KS = dim / 32;
    for (uint32_t i = tid; i < KS * 64; i += blockDim.x) {
        const uint32_t lane = i & 63, ks = i >> 6;
        const uint32_t ql = lane & 31;
        const float* q = qpad + (size_t)(c * 32 + ql) * dim + ks * 32 + 16 * (lane >> 5);
        const float sc = sq[ql];
        uint32_t w[4];
        for (int e = 0; e < 4; ++e) {
            uint32_t word = 0;
            for (int b = 0; b < 4; ++b) {
                int Q = __float2int_rn(q[4 * e + b] / sc);
                Q = Q > 127 ? 127 : (Q < -127 ? -127 : Q);
                word |= (uint32_t)(Q & 0xff) << (8 * b);
            }
            w[e] = word;
        }
        qb8[(c * KS + ks) * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// integer screen threshold: T = floor((thr - eps) / (s_x s_q)) - 1 as int32 bits (INT_MIN: everything passes)
__global__ void screen_thr8_kernel(const float* __restrict__ thr, const float* __restrict__ eps,
                                   const float* __restrict__ qscale, float s_x, float* __restrict__ thr_screen) {
    const uint32_t q = threadIdx.x;
    if (q >= (uint32_t)kMaxQueries) return;
    const float t = thr[q], e = eps[q];
    int T = (int)0x80000000;
    if (e == e && e <= 1e30f && t == t && t > -__builtin_inff()) {
        const double v = floor(((double)t - (double)e) / ((double)s_x * (double)qscale[q])) - 1.0;
        T = v >= 2147483647.0 ? 0x7fffffff : (v <= -2147483648.0 ? (int)0x80000000 : (int)v);
    }
    thr_screen[q] = __int_as_float(T);
}

// Squared-Euclidean recall on the int8 screen.  A pair can reach the threshold thr (in -d units) only if, in exact arithmetic,
// 2 ip - |x|^2 - |q|^2 >= thr - delta (delta: the rounding of the specification's own fp32 evaluation of -d), i.e.
// ip >= (thr - delta + |q|^2) / 2 + |x|^2 / 2; with ip <= s_x s_q I + eps_q (the inner-product screen's bound) and |x|^2 >= the
// block's smallest: I >= A_q + B_q min|x|^2,  A_q = ((thr - delta + |q|^2) / 2 - eps_q) / (s_x s_q),  B_q = 1 / (2 s_x s_q).
// Both are rounded down here, the kernel rounds the sum down again.  thr = -inf (or anything odd): A_q = -inf, everything passes.
__global__ void screen_thr8_l2_kernel(const float* __restrict__ thr, const float* __restrict__ eps, const float* __restrict__ qscale,
                                      const float* __restrict__ nqv, float s_x, float max_norm, float* __restrict__ a_out,
                                      float* __restrict__ b_out, int per_row) {
    const uint32_t q = threadIdx.x;
    if (q >= (uint32_t)kMaxQueries) return;
    const float t = thr[q], e = eps[q];
    const double nq = (double)nqv[q], N = (double)max_norm;
    const double sq = (double)s_x * (double)qscale[q];
    float A = -__builtin_inff(), B = 0.0f;
    if (e == e && e <= 1e30f && t == t && t > -__builtin_inff() && sq > 0.0 && nq == nq && nq < 1e30) {
        const double delta = 2e-6 * (N * N + nq + 2.0 * N * sqrt(nq)) + 1e-30;
        if (per_row) {
            // per-row form: 2 s_x s_q I - |x|^2 >= thr - delta + |q|^2 - 2 eps_q =: C_q, evaluated by the kernel as
            // fmaf((float)I, c1, -|x|^2) with c1 = 2 s_x s_q rounded UP (I can be negative: the product's error is covered
            // with the fma's by a second delta)
            const double c = (double)t - 2.0 * delta + nq - 2.0 * (double)e;
            A = (float)(c - fabs(c) * 1e-6 - 1e-30);
            B = (float)(2.0 * sq);
            if (!(A == A) || !(B == B) || B > 1e30f) { A = -__builtin_inff(); B = 0.0f; }
        } else {
            const double a = (((double)t - delta + nq) * 0.5 - (double)e) / sq;
            const double b = 0.5 / sq;
            A = (float)(a - fabs(a) * 1e-6 - 1.0);
            B = (float)(b * (1.0 - 1e-6));
            if (!(A == A) || !(B == B) || B > 1e30f) { A = -__builtin_inff(); B = 0.0f; }
        }
    }
    a_out[q] = A;
    b_out[q] = B;
}
// smallest |x|^2 of every 32-row block (rows past the table's end do not count)
// stats[0] += sum over rows of (|x|^2 - the block's smallest), stats[1] += sum of |x|^2: how much the per-block cutoff gives away
__global__ void block_nxmin_kernel(const float* __restrict__ nx, uint64_t rows, float* __restrict__ out, uint32_t nblocks,
                                   double* __restrict__ stats) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    double slack = 0.0, sum = 0.0;
    if (b < nblocks) {
        float m = __builtin_inff();
        for (uint32_t i = 0; i < (uint32_t)kPieceRows; ++i) {
            const uint64_t r = (uint64_t)b * kPieceRows + i;
            if (r < rows) m = fminf(m, nx[r]);
        }
        m = m == m ? m : 0.0f;                       // (a NaN norm: no help from this block)
        out[b] = m;
        for (uint32_t i = 0; i < (uint32_t)kPieceRows; ++i) {
            const uint64_t r = (uint64_t)b * kPieceRows + i;
            if (r < rows && nx[r] == nx[r] && nx[r] < 1e30f) { slack += (double)nx[r] - (double)m; sum += (double)nx[r]; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        slack += __shfl_xor(slack, off, 64);
        sum += __shfl_xor(sum, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && sum > 0.0) {
        atomicAdd(&stats[0], slack);
        atomicAdd(&stats[1], sum);
    }
}

// Table statistics without a shadow (first pass of the int8 build): max |x|, max row L2 norm^2, finiteness.
template <int DIM>
__global__ __launch_bounds__(256) void table_stats_kernel(const float* __restrict__ tab, uint64_t rows,
                                                          float* __restrict__ out_max, uint32_t* __restrict__ out_nonfinite,
                                                          float* __restrict__ out_absmax, float* __restrict__ out_sumsq) {
    constexpr int G = DIM / 8;
    __shared__ float smax[4], samax[4], ssum[4];
    __shared__ uint32_t sbad[4];
    const uint64_t n8 = rows * (uint64_t)G;
    float mx = 0.0f, amx = 0.0f, sum = 0.0f;
    uint32_t bad = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ((n8 + 63) & ~63ull);
         g += (uint64_t)gridDim.x * blockDim.x) {
        float ss = 0.0f;
        if (g < n8) {
            const float4 a = reinterpret_cast<const float4*>(tab)[2 * g];
            const float4 b = reinterpret_cast<const float4*>(tab)[2 * g + 1];
            ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
            sum += ss;
            amx = fmaxf(amx, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                   fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
        }
#pragma unroll
        for (int off = 1; off < G; off <<= 1) ss += __shfl_xor(ss, off, 64);
        if (!(ss < 3.0e38f)) bad = 1;
        mx = fmaxf(mx, ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        amx = fmaxf(amx, __shfl_xor(amx, off, 64));
        sum += __shfl_xor(sum, off, 64);
        bad |= (uint32_t)__shfl_xor((int)bad, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smax[w] = mx; samax[w] = amx; ssum[w] = sum; sbad[w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(reinterpret_cast<uint32_t*>(out_max), __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))));
        atomicMax(reinterpret_cast<uint32_t*>(out_absmax),
                  __float_as_uint(fmaxf(fmaxf(samax[0], samax[1]), fmaxf(samax[2], samax[3]))));
        if (sbad[0] | sbad[1] | sbad[2] | sbad[3]) atomicOr(out_nonfinite, 1u);
        atomicAdd(out_sumsq, ssum[0] + ssum[1] + ssum[2] + ssum[3]);     // (statistics only: decides int8 vs bf16)
    }
}

// Second pass: the int8 shadow X = clamp(rint(x / s), +-127) and the largest row residual ||x - s X||^2.
// A thread converts 8 consecutive values (one 8-byte store); DIM/8 neighbouring lanes share a row.
template <int DIM>
__global__ __launch_bounds__(256) void table_quant8_kernel(const float* __restrict__ tab, uint64_t rows, float s,
                                                           int8_t* __restrict__ out8, float* __restrict__ out_resid) {
    constexpr int G = DIM / 8;
    __shared__ float smax[4];
    const uint64_t n8 = rows * (uint64_t)G;
    const float inv = 1.0f / s;
    float mx = 0.0f;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ((n8 + 63) & ~63ull);
         g += (uint64_t)gridDim.x * blockDim.x) {
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
                rs = __fmaf_rn(r, r, rs);
                w[i >> 2] |= (uint32_t)(X & 0xff) << (8 * (i & 3));
            }
            reinterpret_cast<uint2*>(out8)[g] = make_uint2(w[0], w[1]);
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
        atomicMax(reinterpret_cast<uint32_t*>(out_resid), __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))));
}

// Table preparation for the screen, one coalesced pass: the bf16 shadow of the rows (RNE, what
// v_cvt_pk_bf16_f32 gives) and the statistics the error bound needs — max row L2 norm (upper bound) and
// finiteness.  A thread converts 8 consecutive values; DIM/8 neighbouring lanes share a row.
template <int DIM>
__global__ __launch_bounds__(256) void table_shadow_kernel(const float* __restrict__ tab, uint64_t rows,
                                                           uint16_t* __restrict__ out16,
                                                           float* __restrict__ out_max,
                                                           uint32_t* __restrict__ out_nonfinite,
                                                           float* __restrict__ out_blk_norm2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    constexpr int G = DIM / 8;                       // lanes per row (8 or 16)
    __shared__ float smax[4];
    __shared__ uint32_t sbad[4];
    const uint64_t n8 = rows * (uint64_t)G;          // 8-value groups in the table
    float mx = 0.0f;
    uint32_t bad = 0;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ((n8 + 63) & ~63ull);
         g += (uint64_t)gridDim.x * blockDim.x) {
        float ss = 0.0f;
        if (g < n8) {
            const float4 a = reinterpret_cast<const float4*>(tab)[2 * g];
            const float4 b = reinterpret_cast<const float4*>(tab)[2 * g + 1];
            const f32x2 p0 = {a.x, a.y}, p1 = {a.z, a.w}, p2 = {b.x, b.y}, p3 = {b.z, b.w};
            uint4 o;
            o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(p0, bf16x2));
            o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, bf16x2));
            o.z = __builtin_bit_cast(uint32_t, __builtin_convertvector(p2, bf16x2));
            o.w = __builtin_bit_cast(uint32_t, __builtin_convertvector(p3, bf16x2));
            reinterpret_cast<uint4*>(out16)[g] = o;
            ss = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        }
#pragma unroll
        for (int off = 1; off < G; off <<= 1) ss += __shfl_xor(ss, off, 64);     // the row's sum of squares
        if (!(ss < 3.0e38f)) bad = 1;                 // NaN, inf or overflow
        mx = fmaxf(mx, ss);
        // largest row norm^2 of the 32-row block: a wave holds 64 / G consecutive rows of one block
        float wm = ss;
#pragma unroll
        for (int off = G; off < 64; off <<= 1) wm = fmaxf(wm, __shfl_xor(wm, off, 64));
        if ((threadIdx.x & 63) == 0 && g < n8)
            atomicMax(reinterpret_cast<uint32_t*>(out_blk_norm2) + (g / G) / kPieceRows, __float_as_uint(fmaxf(wm, 0.0f)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        bad |= (uint32_t)__shfl_xor((int)bad, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smax[w] = mx; sbad[w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        atomicMax(reinterpret_cast<uint32_t*>(out_max), __float_as_uint(m));   // non-negative floats order as uints
        if (sbad[0] | sbad[1] | sbad[2] | sbad[3]) atomicOr(out_nonfinite, 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// select: keep the K largest keys of cand_in[q][0..M) in cand_out[q][0..min(M,K)), set cnt, thr.
// One 1024-thread workgroup per query; 8 radix passes (one byte each) over L2-resident keys.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSelLdsKeys = 16384;          // candidate lists up to this size are selected in LDS

// Block-wide radix-select helper state lives in LDS; keys come from `src` (LDS or global).
__global__ __launch_bounds__(1024) void select_kernel(const uint64_t* __restrict__ cand_in,
                                                      uint64_t* __restrict__ cand_out,
                                                      uint32_t* __restrict__ cnt,
                                                      float* __restrict__ thr, uint32_t cap,
                                                      uint32_t K, int thr_only, uint32_t* __restrict__ cnt_seen) {
    // thr_only: the lists stay as they are (cand_out is not written, cnt unchanged); the query's threshold rises to the
    // score of its K-th largest candidate so far — when it has that many (the refinement step of the pilot plan)
    extern __shared__ __attribute__((aligned(16))) char sel_smem[];
    uint64_t* const lkeys = reinterpret_cast<uint64_t*>(sel_smem);          // [kSelLdsKeys] when used
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_digit, s_need, s_out, s_cnt;
    __shared__ unsigned long long s_min, s_max;
    const uint32_t q = blockIdx.x;
    const uint64_t* in = cand_in + (uint64_t)q * cap;
    uint64_t* out = cand_out + (uint64_t)q * cap;
    uint32_t M = cnt[q];
    if (threadIdx.x == 0 && cnt_seen) cnt_seen[q] = M;      // (statistics: the candidates collected before K of them are kept)
    if (M > cap) M = cap;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { s_min = ~0ull; s_max = 0ull; }
    __syncthreads();
    const bool in_lds = M <= kSelLdsKeys;
    // pass 0: stage the keys in LDS (when they fit) and find their range
    unsigned long long mn = ~0ull, mx = 0ull;
    for (uint32_t i = tid; i < M; i += 1024) {
        const uint64_t k = in[i];
        if (in_lds) lkeys[i] = k;
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
    }
    // wave-level reduce, then one LDS atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long omn = __shfl_xor(mn, off), omx = __shfl_xor(mx, off);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
    }
    if ((tid & 63) == 0) {
        atomicMin(&s_min, mn);
        atomicMax(&s_max, mx);
    }
    __syncthreads();
    const uint64_t kmin = s_min, kmax = s_max;
    const uint64_t* src = in_lds ? lkeys : in;
    if (M <= K) {
        if (!thr_only)
            for (uint32_t i = tid; i < M; i += 1024) out[i] = src[i];
        if (tid == 0) {
            if (!thr_only) cnt[q] = M;
            if (M == K && K > 0) thr[q] = key_score(kmin);     // threshold = score of the smallest key
        }
        return;
    }
    // radix select of the K-th largest of (key - kmin): only the bits below the span's top bit vary,
    // so the first digit is already well spread (no single hot histogram bin)
    const uint64_t span = kmax - kmin;
    int shift = span ? (63 - __clzll((long long)span)) - 7 : 0;
    if (shift < 0) shift = 0;
    uint64_t prefix = 0, mask = 0;              // over (key - kmin)
    uint32_t need = K;
    uint32_t bucket = M;
    while (true) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < M; i += 1024) {
            const uint64_t k = src[i] - kmin;
            if ((k & mask) == prefix) atomicAdd(&hist[(uint32_t)(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t above = 0;
            for (int d = 255; d > (int)tid; --d) above += hist[d];
            if (above < need && need <= above + hist[tid]) {
                s_digit = tid;
                s_need = need - above;
                s_cnt = hist[tid];
            }
        }
        __syncthreads();
        prefix |= (uint64_t)s_digit << shift;
        mask |= 255ull << shift;
        need = s_need;
        bucket = s_cnt;
        __syncthreads();
        if (shift == 0 || bucket == 1) break;
        shift = shift >= 8 ? shift - 8 : 0;
    }
    // keys are distinct.  If the selected bucket holds one key, that key is the K-th; otherwise all
    // digits down to bit 0 were fixed and prefix is the full (key - kmin) of the K-th.
    if (shift != 0) {
        if (tid == 0) s_min = 0ull;
        __syncthreads();
        for (uint32_t i = tid; i < M; i += 1024) {
            const uint64_t k = src[i] - kmin;
            if ((k & mask) == prefix) s_min = k;            // exactly one thread writes
        }
        __syncthreads();
        prefix = s_min;
    }
    const uint64_t kth = prefix + kmin;          // exactly K keys are >= kth
    if (thr_only) {
        if (tid == 0) thr[q] = key_score(kth);
        return;
    }
    if (tid == 0) s_out = 0;
    __syncthreads();
    for (uint32_t i = tid; i < M; i += 1024) {
        const uint64_t k = src[i];
        if (k >= kth) out[atomicAdd(&s_out, 1u)] = k;
    }
    if (tid == 0) {
        cnt[q] = K;
        thr[q] = key_score(kth);
    }
}

// final: sort the survivors descending in LDS and decode.  P = pow2 >= n, P*8 bytes of LDS.
__global__ __launch_bounds__(1024) void final_kernel(const uint64_t* __restrict__ cand,
                                                     const uint32_t* __restrict__ cnt, uint32_t cap,
                                                     uint32_t K, uint32_t P, uint64_t row_offset,
                                                     uint64_t* __restrict__ out_rows,
                                                     float* __restrict__ out_scores,
                                                     uint32_t* __restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    uint32_t n = cnt[q];
    if (n > K) n = K;
    const uint64_t* in = cand + (uint64_t)q * cap;
    for (uint32_t i = tid; i < P; i += 1024) s[i] = i < n ? in[i] : 0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t idx = tid; idx < P; idx += 1024) {
                const uint32_t ixj = idx ^ j;
                if (ixj > idx) {
                    const uint64_t x = s[idx], y = s[ixj];
                    const bool desc = (idx & k) == 0;
                    if ((x < y) == desc) {
                        s[idx] = y;
                        s[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < K; i += 1024) {
        if (i < n) {
            out_rows[(uint64_t)q * K + i] = row_offset + key_row(s[i]);
            out_scores[(uint64_t)q * K + i] = key_score(s[i]);
        } else {
            out_rows[(uint64_t)q * K + i] = ~0ull;
            out_scores[(uint64_t)q * K + i] = -__builtin_inff();
        }
    }
    if (tid == 0 && out_count) out_count[q] = n;
}

// final for K <= 8192: the register-resident bitonic network (bitonic_reg.hpp) on complemented keys
__global__ __launch_bounds__(1024) void final_kernel_reg(const uint64_t* __restrict__ cand,
                                                         const uint32_t* __restrict__ cnt, uint32_t cap,
                                                         uint32_t K, uint64_t row_offset,
                                                         uint64_t* __restrict__ out_rows,
                                                         float* __restrict__ out_scores,
                                                         uint32_t* __restrict__ out_count) {
    __shared__ BitonicLds lds;
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    uint32_t n = cnt[q];
    if (n > K) n = K;
    uint32_t P = 512;
    while (P < K) P <<= 1;
    const uint64_t* in = cand + (uint64_t)q * cap;
    uint64_t k[kBitonicE];
    uint32_t ix[kBitonicE];
#pragma unroll
    for (int u = 0; u < kBitonicE; ++u) {
        const uint32_t i = t * kBitonicE + u;
        k[u] = (i < n) ? ~in[i] : ~0ull;               // descending = ascending on the complement; pad last
        ix[u] = 0;
    }
    bitonic_sort_reg<false>(k, ix, P, lds, n);
#pragma unroll
    for (int u = 0; u < kBitonicE; ++u) {
        const uint32_t i = t * kBitonicE + u;
        if (i < K) {
            const uint64_t key = ~k[u];
            if (i < n) {
                out_rows[(uint64_t)q * K + i] = row_offset + key_row(key);
                out_scores[(uint64_t)q * K + i] = key_score(key);
            } else {
                out_rows[(uint64_t)q * K + i] = ~0ull;
                out_scores[(uint64_t)q * K + i] = -__builtin_inff();
            }
        }
    }
    if (t == 0 && out_count) out_count[q] = n;
}

// final for a handful of queries (K <= 8192): counting ranks instead of a sorting network.  One workgroup sorts a
// query's 8192-slot network on ONE CU in 48 us whatever the chip is doing; with only a few queries in the call the
// other 250 CUs are idle, so every 64 candidates get a workgroup of their own: it stages the query's n keys in LDS
// (broadcast reads), wave p counts the keys above each of its 64 within the p-th quarter, the four counts add up
// to the candidate's rank — its output position, the keys being distinct.  n^2 / 4 compares per wave: ~6 us at 5 000.
// EPB candidates per workgroup, 256 / EPB threads each: 16 for one or two queries, 64 beyond.
template <int EPB>
__global__ __launch_bounds__(256) void final_rank_kernel(const uint64_t* __restrict__ cand, const uint32_t* __restrict__ cnt,
                                                         uint32_t cap, uint32_t K, uint64_t row_offset,
                                                         uint64_t* __restrict__ out_rows, float* __restrict__ out_scores,
                                                         uint32_t* __restrict__ out_count) {
    constexpr uint32_t PARTS = 256 / EPB;
    extern __shared__ __attribute__((aligned(16))) uint64_t rk_keys[];       // [n rounded up to 32]
    __shared__ uint32_t part[PARTS][EPB];
    const uint32_t q = blockIdx.y, tid = threadIdx.x;
    uint32_t n = cnt[q];
    if (n > K) n = K;
    // positions past the candidates (fewer than K rows in the table)
    for (uint32_t i = n + blockIdx.x * 256u + tid; i < K; i += gridDim.x * 256u) {
        out_rows[(uint64_t)q * K + i] = ~0ull;
        out_scores[(uint64_t)q * K + i] = -__builtin_inff();
    }
    if (blockIdx.x == 0 && tid == 0 && out_count) out_count[q] = n;
    if (blockIdx.x * (uint32_t)EPB >= n) return;
    const uint64_t* in = cand + (uint64_t)q * cap;
    const uint32_t n8 = (n + 31u) & ~31u;
    for (uint32_t i0 = tid; i0 < n8; i0 += 8u * 256u) {                     // eight independent loads per thread in flight
        uint64_t kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + (uint32_t)u * 256u;
            kv[u] = in[i < n ? i : 0u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + (uint32_t)u * 256u;
            if (i < n8) rk_keys[i] = i < n ? kv[u] : 0ull;                  // padding: below every real key
        }
    }
    __syncthreads();
    const uint32_t el = tid % (uint32_t)EPB, p = tid / (uint32_t)EPB;
    const uint32_t e = blockIdx.x * (uint32_t)EPB + el;
    const uint64_t mine = e < n ? rk_keys[e] : ~0ull;
    const uint32_t chunk = n8 / PARTS;                                     // n8 % 32 == 0: even chunks
    const uint32_t j0 = p * chunk, j1 = j0 + chunk;
    uint32_t above = 0;
    // (unrolled: one wave per SIMD here, so a broadcast read's LDS latency is hidden only by the reads behind it)
#pragma unroll 4
    for (uint32_t j = j0; j < j1; j += 2) {
        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&rk_keys[j]);
        above += (kk.x > mine) ? 1u : 0u;
        above += (kk.y > mine) ? 1u : 0u;
    }
    part[p][el] = above;
    __syncthreads();
    if (p == 0 && e < n) {
        uint32_t r = 0;
#pragma unroll
        for (uint32_t i = 0; i < PARTS; ++i) r += part[i][el];
        out_rows[(uint64_t)q * K + r] = row_offset + key_row(mine);
        out_scores[(uint64_t)q * K + r] = key_score(mine);
    }
}

// after a refined pilot plan: thr = score of the K-th best candidate kept (select_kernel), thr_ref = the raised threshold
__global__ void refine_verify_kernel(const float* __restrict__ thr, const float* __restrict__ thr_ref,
                                     uint32_t* __restrict__ count, uint32_t nq) {
    const uint32_t q = threadIdx.x;
    if (q < nq && !(thr[q] >= thr_ref[q])) count[q] = 0u;
}

__global__ void recall_init_kernel(const float* __restrict__ queries, uint32_t nq, uint32_t dim,
                                   float* __restrict__ qpad, float* __restrict__ thr,
                                   uint32_t* __restrict__ cnt, uint32_t* __restrict__ overflow) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < kMaxQueries * dim) qpad[i] = (i / dim) < nq ? queries[i] : 0.0f;
    if (i < kMaxQueries) {
        thr[i] = -__builtin_inff();
        cnt[i] = 0;
        cnt[kMaxQueries + i] = 0;                          // susp_cnt (RecallScratch: right behind cnt)
    }
    if (i == 0) { *overflow = 0; overflow[kRecOvfWord] = 0; }
}

// squared-Euclidean recall: |x|^2 of every row and |q|^2 of every query as k-ascending fmaf chains (the specification's), and
// the sign flip of the scores that come out (-d → d; padding -inf → +inf)
__global__ void row_norm2_kernel(const float* __restrict__ tab, uint64_t rows, uint32_t dim, float* __restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float4* x = reinterpret_cast<const float4*>(tab + r * dim);
    float s = 0.0f;
    for (uint32_t c = 0; c < dim / 4; ++c) {
        const float4 v = x[c];
        s = __fmaf_rn(v.x, v.x, s);
        s = __fmaf_rn(v.y, v.y, s);
        s = __fmaf_rn(v.z, v.z, s);
        s = __fmaf_rn(v.w, v.w, s);
    }
    out[r] = s;
}
__global__ void query_norm2_kernel(const float* __restrict__ qpad, uint32_t dim, float* __restrict__ out) {
    const uint32_t q = threadIdx.x;
    float s = 0.0f;
    for (uint32_t c = 0; c < dim; ++c) s = __fmaf_rn(qpad[(size_t)q * dim + c], qpad[(size_t)q * dim + c], s);
    out[q] = s;
}
__global__ void negate_kernel(float* __restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = 0.0f - v[i];                     // (a zero distance comes out as +0, padding as +inf)
}

// merge input lists (global rows, scores) → candidate keys.  Padding entries (row = UINT64_MAX: a shard with fewer
// than per_list rows) are dropped here, so the keys stay distinct (select_kernel's invariant) and cnt[q] is the number
// of real candidates.  Input layout: [nq][nlists][per_list] (list_major = 0) or [nlists][nq][per_list] (1: what an
// all-gather of per-shard [nq][per_list] blocks produces).  cnt must be zero on entry.
// list l of query q starts at rows + l * row_ls + q * row_qs (scores likewise): covers the query-major layout
// [nq][nlists][per_list], the list-major one an all-gather produces, and packed per-shard (rows | scores) blocks
__global__ void merge_keys_kernel(const uint64_t* __restrict__ rows, const float* __restrict__ scores, uint32_t nq,
                                  uint32_t nlists, uint32_t per_list, size_t row_ls, size_t row_qs, size_t sc_ls, size_t sc_qs,
                                  uint32_t cap, uint64_t* __restrict__ cand, uint32_t* __restrict__ cnt) {
    const uint32_t q = blockIdx.y;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per_q = nlists * per_list;
    if (i >= per_q) return;
    const uint32_t l = i / per_list, j = i - l * per_list;
    const uint64_t r = rows[l * row_ls + q * row_qs + j];
    if (r == ~0ull) return;
    cand[(uint64_t)q * cap + atomicAdd(&cnt[q], 1u)] = topk_key(scores[l * sc_ls + q * sc_qs + j], (uint32_t)r);
}

// ---------------------------------------------------------------------------------------------
// Threshold predictor.  A query's scores over the table are a projection of the rows: mean mu.q, variance q'Sq, and —
// 128 terms each — close to Gaussian, so the K-th best of N rows sits z standard deviations above the mean, with z
// nearly the same for every query of a workload.  mu and S come from a row sample (with the table's statistics); z is
// not assumed but OBSERVED: every verified batch reports (K-th best − mu.q) / sigma_q of its queries.  Once the
// observed z is tight, the first thresholds of a batch are mu.q + (mean z − margin) sigma_q: the pilot sample — two
// small scan launches, a re-scoring and two selects, 0.35 ms that nothing overlaps — is skipped.  Exactness does
// not depend on any of this: a threshold that turns out too high leaves a query short of K candidates, which the
// plan check sees, and the batch re-runs on the pilot plan (and the table stays on it for a while).
// ---------------------------------------------------------------------------------------------
// second moments of a row sample: workgroup b walks sample rows b, b + G, ...; thread (ty, tx) of a 16 x 16 grid owns
// the 8 x 8 patch (i = ty + 16a, j = tx + 16b) of sum x_i x_j; 16 rows are staged per barrier pair
__global__ __launch_bounds__(256) void pred_moments_kernel(const float* __restrict__ tab, uint64_t rows, uint64_t stride,
                                                           uint64_t n_sample, float* __restrict__ sum1, float* __restrict__ sum2) {
    __shared__ float xs[16][128 + 4];
    const uint32_t tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    float acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.0f;
    float s1 = 0.0f;                                       // thread t < 128: sum of column t
    for (uint64_t r0 = (uint64_t)blockIdx.x * 16; r0 < n_sample; r0 += (uint64_t)gridDim.x * 16) {
        __syncthreads();
        for (uint32_t e = tid; e < 16 * 32; e += 256) {    // 16 rows x 32 quads
            const uint32_t rr = e >> 5, qd = e & 31;
            const uint64_t sr = r0 + rr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sr < n_sample) {
                const uint64_t row = sr * stride < rows ? sr * stride : rows - 1;
                v = *reinterpret_cast<const float4*>(tab + row * 128 + 4 * qd);
            }
            *reinterpret_cast<float4*>(&xs[rr][4 * qd]) = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {
            float xi[8], xj[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                xi[a] = xs[rr][ty + 16 * a];
                xj[a] = xs[rr][tx + 16 * a];
            }
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b) acc[a][b] = __fmaf_rn(xi[a], xj[b], acc[a][b]);
            if (tid < 128) s1 += xs[rr][tid];
        }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) atomicAdd(&sum2[(ty + 16 * a) * 128 + tx + 16 * b], acc[a][b]);
    if (tid < 128) atomicAdd(&sum1[tid], s1);
}
// sums → mean and covariance in place; the observation block behind them is cleared
__global__ void pred_finish_kernel(float* __restrict__ pred, uint64_t n_sample) {
    const uint32_t i = blockIdx.x, j = threadIdx.x;        // 128 x 128
    const float inv = 1.0f / (float)n_sample;
    const float mi = pred[i] * inv, mj = pred[j] * inv;
    const float c = pred[128 + i * 128 + j] * inv - mi * mj;
    pred[128 + i * 128 + j] = c;                           // (the raw column sums pred[0..127] are turned into means by the next launch)
}
__global__ void pred_means_kernel(float* __restrict__ pred, uint64_t n_sample) {
    const uint32_t j = threadIdx.x;
    if (j < 128) pred[j] = pred[j] / (float)n_sample;
    if (j < 8) reinterpret_cast<uint32_t*>(pred + 128 + 128 * 128)[j] = 0u;          // n, sum z, sum z^2, min z (doubles)
    if (j == 0) reinterpret_cast<double*>(pred + 128 + 128 * 128)[3] = 1e300;
}
// per query: mean mu.q and sigma sqrt(q'Sq) of its scores (one workgroup of 128 threads per query)
__global__ __launch_bounds__(128) void pred_query_kernel(const float* __restrict__ qpad, const float* __restrict__ pred,
                                                         float* __restrict__ ms, float* __restrict__ thr, float z_lo) {
    __shared__ float qs[128];
    __shared__ double red[2][2];
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    qs[t] = qpad[(size_t)q * 128 + t];
    __syncthreads();
    const float* S = pred + 128;
    float sq = 0.0f;
    for (int i = 0; i < 128; ++i) sq = __fmaf_rn(S[i * 128 + t], qs[i], sq);          // (S q)_t: S symmetric, column t coalesced
    double v = (double)sq * (double)qs[t], m = (double)pred[t] * (double)qs[t];
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_xor(v, off, 64);
        m += __shfl_xor(m, off, 64);
    }
    if ((t & 63) == 0) { red[t >> 6][0] = v; red[t >> 6][1] = m; }
    __syncthreads();
    if (t == 0) {
        const double var = red[0][0] + red[1][0], mean = red[0][1] + red[1][1];
        const float m = (float)mean, sg = var > 0.0 ? (float)sqrt(var) : 0.0f;
        ms[2 * q] = m;
        ms[2 * q + 1] = sg;
        if (thr) {
            // first threshold from the model: mean + z_lo sigma, rounded down; anything odd → +inf, which leaves the query
            // without candidates: the plan check then sends the batch to the pilot plan.  (Query columns >= nq keep the
            // -inf recall_init_kernel gave them.)
            float t = __fmaf_rn(z_lo, sg, m);
            t = t - fabsf(t) * 1e-6f;
            if (!(sg > 0.0f) || !(t == t) || fabsf(t) > 1e30f) t = __builtin_inff();
            thr[q] = t;
        }
    }
}
// after a verified-to-be pass: fold the batch's observed quantiles into the table's statistics (queries that came up
// short of K, or whose model sigma is 0, do not count), then they travel to the host with the status words
__global__ void pred_update_kernel(const float* __restrict__ thr, const float* __restrict__ ms, const uint32_t* __restrict__ cnt,
                                   uint32_t nq, uint32_t k, double* __restrict__ stats) {
    const uint32_t q = threadIdx.x;
    double z = 0.0, z2 = 0.0, n = 0.0, mn = 1e300;
    if (q < nq && cnt[q] >= k && ms[2 * q + 1] > 0.0f) {
        const float t = thr[q];
        if (t == t && fabsf(t) < 1e30f) {
            z = ((double)t - (double)ms[2 * q]) / (double)ms[2 * q + 1];
            z2 = z * z;
            n = 1.0;
            mn = z;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        z += __shfl_xor(z, off, 64);
        z2 += __shfl_xor(z2, off, 64);
        n += __shfl_xor(n, off, 64);
        const double o = __shfl_xor(mn, off, 64);
        mn = o < mn ? o : mn;
    }
    if ((q & 63) == 0 && n > 0.0) {
        atomicAdd(&stats[4], n);                       // observations ever (orders the host's snapshots; never decays)
        atomicAdd(&stats[0], n);
        atomicAdd(&stats[1], z);
        atomicAdd(&stats[2], z2);
        // (min via compare-and-swap on the bit pattern: the values are positive in every workload this is for, and a
        //  stale minimum only makes the margin more careful)
        unsigned long long* a = reinterpret_cast<unsigned long long*>(&stats[3]);
        unsigned long long old = *a;
        while (__longlong_as_double((long long)old) > mn) {
            const unsigned long long seen = atomicCAS(a, old, (unsigned long long)__double_as_longlong(mn));
            if (seen == old) break;
            old = seen;
        }
    }
    // a sliding window in effect: past 2^17 observations the sums halve, so a drifting query population moves the
    // model within ~100 batches (the halving may race another stream's add and drop it: these are statistics)
    __syncthreads();
    if (q == 0 && stats[0] >= 131072.0) {
        stats[0] *= 0.5;
        stats[1] *= 0.5;
        stats[2] *= 0.5;
    }
}

int launch_select(pg_ctx* ctx, uint32_t nq, const uint64_t* in, uint64_t* out, uint32_t* cnt, float* thr,
                         uint32_t cap, uint32_t k, int thr_only, uint32_t* cnt_seen) {
    constexpr size_t lds = (size_t)kSelLdsKeys * 8;
    int rc_attr;
    if ((rc_attr = ensure_dyn_lds(ctx, (const void*)select_kernel, lds))) return rc_attr;
    select_kernel<<<nq, 1024, lds, ctx->stream>>>(in, out, cnt, thr, cap, k, thr_only, cnt_seen);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

static uint32_t next_pow2(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

template <int DIM, int NQB = 1, int VAR = 0, bool L2 = false>
static int launch_scan(pg_ctx* ctx, const ScanArgs& a) {
    int rc_attr;
    if ((rc_attr = ensure_dyn_lds(ctx, (const void*)scan_kernel<DIM, NQB, VAR, L2>, kScanLds))) return rc_attr;
    const uint32_t total = a.rb_end - a.rb_begin;
    uint32_t grid = (uint32_t)ctx->num_cus;
    const uint32_t need = (total + kScanWaves - 1) / kScanWaves;
    if (grid > need) grid = need;
    const uint32_t groups = a.group_q ? (a.nq + a.group_q - 1) / a.group_q : 1;
    scan_kernel<DIM, NQB, VAR, L2><<<dim3(grid, groups), 64 * kScanWaves, kScanLds, ctx->stream>>>(a);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

static int dispatch_scan(pg_ctx* ctx, uint32_t dim, const ScanArgs& a) {
    const bool wide = a.nq_launch > 32;          // two 32-query column blocks
    if (a.nx) {                                   // squared-Euclidean recall
        if (dim == 64) return wide ? launch_scan<64, 2, 0, true>(ctx, a) : launch_scan<64, 1, 0, true>(ctx, a);
        if (dim == 128) return wide ? launch_scan<128, 2, 0, true>(ctx, a) : launch_scan<128, 1, 0, true>(ctx, a);
        set_error("recall (squared Euclidean): dim=%u unsupported (64 or 128)", dim);
        return PG_ERR_UNSUPPORTED;
    }
    switch (dim) {
        case 64: return wide ? launch_scan<64, 2>(ctx, a) : launch_scan<64, 1>(ctx, a);
        case 128: {
#ifdef PG_SCAN_VARIANTS
            const char* v = getenv("PG_SCAN_VAR");     // developer ablation builds only
            if (v && v[0] == '1') return launch_scan<128, 1, 1>(ctx, a);
            if (v && v[0] == '2') return launch_scan<128, 1, 2>(ctx, a);
            if (v && v[0] == '3') return launch_scan<128, 1, 3>(ctx, a);
#endif
            return wide ? launch_scan<128, 2>(ctx, a) : launch_scan<128, 1>(ctx, a);
        }
        case 192:
        case 256:
            if (wide) {      // 2 x dim/2 B-operand registers no longer fit 2 waves per SIMD
                set_error("recall: dim=%u supports at most 32 queries per pass (64 up to dim 128)", dim);
                return PG_ERR_UNSUPPORTED;
            }
            return dim == 192 ? launch_scan<192, 1>(ctx, a) : launch_scan<256, 1>(ctx, a);
    }
    set_error("recall: dim=%u unsupported", dim);
    return PG_ERR_UNSUPPORTED;
}

constexpr uint32_t kFirstChunkRows = 32768;
constexpr uint32_t kRescoreBlocksPerQuery = 16;   // x 256 suspects per block per stride step
constexpr uint32_t kCandSlack = 1u << 19;      // candidate capacity beyond K per query

int recall_scratch(pg_ctx* ctx, uint32_t dim, uint32_t k, RecallScratch* rs) {
    const uint32_t cap = k + kCandSlack;
    void* small;
    int rc;
    const size_t qb16_bytes = (size_t)kScreenMaxNQB * (dim / 16) * 64 * 16;
    const size_t small_bytes = (size_t)kMaxQueries * dim * 4 + qb16_bytes + (size_t)kMaxQueries * 40 + 2048 + (size_t)kQ4mWords * 4 + 256 + 64 +
                               (size_t)kMaxQueries * (2 * 128 + 16 + 4) + 192 + (size_t)(kRecallStatusWords + kMaxQueries) * 4 + 64;
    if ((rc = scratch_reserve(ctx, 2, small_bytes, &small))) return rc;
    rs->qpad = (float*)small;
    rs->qb16 = (uint4*)((char*)small + (size_t)kMaxQueries * dim * 4);
    rs->thr = (float*)((char*)rs->qb16 + qb16_bytes);
    rs->eps = rs->thr + kMaxQueries;
    rs->thr_screen = rs->eps + kMaxQueries;
    rs->cnt = (uint32_t*)(rs->thr_screen + kMaxQueries);
    rs->susp_cnt = rs->cnt + kMaxQueries;
    rs->overflow = rs->susp_cnt + kMaxQueries;
    rs->qscale = (float*)(rs->overflow + 64 + kMaxQueries);      // overflow word, the valid counts [kMaxQueries] right behind it (+ 1), padding
    rs->q4 = (uint32_t*)(rs->qscale + kMaxQueries);      // 4 x 128 B + 4 x 16 B
    rs->thr_ref = (float*)(rs->q4 + 160);                // [kMaxQueries]
    rs->pred_ms = rs->thr_ref + kMaxQueries;             // [kMaxQueries][2]
    rs->susp2_cnt = (uint32_t*)(rs->pred_ms + 2 * kMaxQueries);                       // [kI4mMaxQueries] + two statistics words
    rs->q4m = (uint32_t*)(((uintptr_t)(rs->susp2_cnt + kI4mMaxQueries + 2) + 63) & ~(uintptr_t)63);   // (16-byte fragment loads)
    rs->q16 = (uint32_t*)(((uintptr_t)(rs->q4m + kQ4mWords) + 63) & ~(uintptr_t)63);              // [256][32] Qh | [256][32] Ql | [256][4]
    rs->susp2w_cnt = rs->q16 + (size_t)kMaxQueries * (2 * 32 + 4);
    rs->status = (uint32_t*)(((uintptr_t)(rs->susp2w_cnt + kMaxQueries + 2) + 63) & ~(uintptr_t)63);
    rs->cnt_seen = rs->status + kRecallStatusWords;
    void* c;
    if ((rc = scratch_reserve(ctx, 3, (size_t)kMaxQueries * cap * (2 * 8 + 4) + (size_t)kMaxQueries * cap * 4, &c))) return rc;
    rs->cand[0] = (uint64_t*)c;
    rs->cand[1] = rs->cand[0] + (size_t)kMaxQueries * cap;
    rs->susp = (uint32_t*)(rs->cand[1] + (size_t)kMaxQueries * cap);
    rs->susp2 = rs->susp + (size_t)kMaxQueries * cap;
    rs->cap = cap;
    return PG_OK;
}

// final as a split sort (split_sort.hpp): descending by candidate key = ascending on the complement, keys distinct
struct FinalSortPolicy {
    static constexpr bool kWithIdx = false;
    const uint64_t* cand;
    const uint32_t* cnt;
    uint32_t cap, K;
    uint64_t row_offset;
    uint64_t* out_rows;
    float* out_scores;
    uint32_t* out_count;
    __device__ uint32_t count(uint32_t q) const { const uint32_t n = cnt[q]; return n < K ? n : K; }
    __device__ uint64_t key(uint32_t q, uint32_t i) const { return ~cand[(uint64_t)q * cap + i]; }
    __device__ void store(uint32_t q, uint32_t rank, uint64_t k, uint32_t) const {
        out_rows[(uint64_t)q * K + rank] = row_offset + key_row(~k);
        out_scores[(uint64_t)q * K + rank] = key_score(~k);
    }
    // positions past the candidates (fewer than K rows in the table); t = this thread among nt of the list's workgroups
    __device__ void tail(uint32_t q, uint32_t n, uint32_t t, uint32_t nt) const {
        for (uint32_t i = n + t; i < K; i += nt) {
            out_rows[(uint64_t)q * K + i] = ~0ull;
            out_scores[(uint64_t)q * K + i] = -__builtin_inff();
        }
        if (t == 0 && out_count) out_count[q] = n;
    }
};

int final_launch(pg_ctx* ctx, const uint64_t* cand, const uint32_t* cnt, uint32_t cap,
                        uint32_t nq, uint32_t k, uint64_t row_offset, uint64_t* d_out_rows,
                        float* d_out_scores, uint32_t* d_out_count) {
    if (!rank_sort_applies(ctx, nq, k, 1.5) && split_sort_applies(ctx, nq, k))
        return split_sort_launch(ctx, FinalSortPolicy{cand, cnt, cap, k, row_offset, d_out_rows, d_out_scores, d_out_count}, nq, k);
    if (rank_sort_applies(ctx, nq, k, 1.5)) {
        const size_t lds = (size_t)((k + 31u) & ~31u) * 8;
        int rc_attr;
        if (nq <= 2) {
            if ((rc_attr = ensure_dyn_lds(ctx, (const void*)final_rank_kernel<16>, lds))) return rc_attr;
            final_rank_kernel<16><<<dim3((k + 15) / 16, nq), 256, lds, ctx->stream>>>(cand, cnt, cap, k, row_offset, d_out_rows,
                                                                                      d_out_scores, d_out_count);
        } else {
            if ((rc_attr = ensure_dyn_lds(ctx, (const void*)final_rank_kernel<64>, lds))) return rc_attr;
            final_rank_kernel<64><<<dim3((k + 63) / 64, nq), 256, lds, ctx->stream>>>(cand, cnt, cap, k, row_offset, d_out_rows,
                                                                                      d_out_scores, d_out_count);
        }
        PG_HIP(hipGetLastError());
        return PG_OK;
    }
    if (k <= kBitonicMax) {
        final_kernel_reg<<<nq, 1024, 0, ctx->stream>>>(cand, cnt, cap, k, row_offset, d_out_rows, d_out_scores,
                                                       d_out_count);
        PG_HIP(hipGetLastError());
        return PG_OK;
    }
    const uint32_t P = next_pow2(k < 2 ? 2 : k);
    const size_t lds = (size_t)P * 8;
    int rc_attr;
    if ((rc_attr = ensure_dyn_lds(ctx, (const void*)final_kernel, lds))) return rc_attr;
    final_kernel<<<nq, 1024, lds, ctx->stream>>>(cand, cnt, cap, k, P, row_offset, d_out_rows,
                                                 d_out_scores, d_out_count);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// statistics + shadow of a table (lazily, cached until the next upload / fill): int8 for dim 128 (two passes:
// statistics, then quantisation with the table's scale), bf16 for dim 64 or when PG_SCREEN_BF16 is set (A/B runs).
// Tables the screen cannot serve (other dims, no memory for the shadow) keep stats_valid = false.
static std::mutex g_stats_build_mu;   // a table is shared by the contexts of a device (a coalescer's sibling): one of them builds
int ensure_table_stats(pg_ctx* ctx, const pg_table* tc) {
    pg_table* t = const_cast<pg_table*>(tc);          // lazily computed cache
    std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
    if (t->stats_valid || t->shadow_failed) return PG_OK;
    t->i4_ok = t->i4_failed = false;                  // the 4-bit shadow (recall_i4.hip) follows the rows too
    t->i4m_pairs = 0.0f;
    t->rec_scale = 0;
    t->r2_ok = t->r2_failed = false;                  // ... and the residual shadow (recall_r2.hip)
    t->wide_susp = 0.0f;
    t->pred_model = false;                            // ... and the threshold model
    t->prefix_failures = 0;
    if (t->dim != 64 && t->dim != 128) { t->shadow_failed = true; return PG_OK; }
    const bool force_bf16 = ctx->knobs.screen_bf16;
    bool i8 = t->dim == 128 && !force_bf16;
    void* p;
    int rc;
    if ((rc = scratch_reserve(ctx, 4, 4096, &p))) return rc;
    float* d_max = (float*)p + 300;                   // [300] max norm^2, [301] non-finite flag, [302] max |x|, [303] max residual^2, [304] sum of norm^2
    uint32_t* d_bad = (uint32_t*)p + 301;
    PG_HIP(hipMemsetAsync(d_max, 0, 20, ctx->stream));
    const uint32_t grid = (uint32_t)ctx->num_cus * 16;
    if (i8) {
        // statistics first: they decide between the two shadows
        table_stats_kernel<128><<<grid, 256, 0, ctx->stream>>>(t->d, t->rows, d_max, d_bad, d_max + 2, d_max + 4);
        PG_HIP(hipGetLastError());
        PG_HIP(hipMemcpyAsync(ctx->h_status + 300, d_max, 20, hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(hipStreamSynchronize(ctx->stream));
        float mx, amx, sumsq;
        memcpy(&mx, ctx->h_status + 300, 4);
        memcpy(&amx, ctx->h_status + 302, 4);
        memcpy(&sumsq, ctx->h_status + 304, 4);
        t->all_finite = ctx->h_status[301] == 0;
        t->max_norm = sqrtf(mx) * 1.0001f;
        t->s8 = fmaxf(amx / 127.0f, 1e-30f);
        t->resid8 = 0.0f;
        // One scale for the table only works when the largest element is not far above the typical row: the int8
        // margin is ~ s8 sqrt(dim / 12) per unit of ||q|| for EVERY row, the bf16 margin 0.008 x the row's own norm.
        // Heavy tails or outliers (a Student-t table: int8 margin 130 x the bf16 one, every row a suspect, 560 ms per
        // recall instead of 2.4) go to the bf16 shadow with per-block norms.
        const float rms_norm = sqrtf(sumsq / (float)(t->rows ? t->rows : 1));
        const bool force_i8 = ctx->knobs.screen_i8;
        if (t->all_finite && !force_i8 && t->s8 * sqrtf((float)t->dim / 12.0f) > 4.0f * kScreenEps * rms_norm) i8 = false;
    }
    if (i8) {
        if (!t->d8) {
            const size_t bytes = (t->rows + 64) * (size_t)t->dim;
            if (hipMalloc((void**)&t->d8, bytes) != hipSuccess) {
                (void)hipGetLastError();
                t->d8 = nullptr;
                t->shadow_failed = true;              // stay on the exact scan
                return PG_OK;
            }
            PG_HIP(hipMemsetAsync(t->d8 + t->rows * (size_t)t->dim, 0, 64 * (size_t)t->dim, ctx->stream));
        }
        if (t->all_finite) {
            table_quant8_kernel<128><<<grid, 256, 0, ctx->stream>>>(t->d, t->rows, t->s8, t->d8, d_max + 3);
            PG_HIP(hipGetLastError());
            PG_HIP(hipMemcpyAsync(ctx->h_status + 303, d_max + 3, 4, hipMemcpyDeviceToHost, ctx->stream));
            PG_HIP(hipStreamSynchronize(ctx->stream));
            float r2;
            memcpy(&r2, ctx->h_status + 303, 4);
            // (the residuals were accumulated in fp32: a relative 1e-3 and an absolute 1e-6 N cover that)
            t->resid8 = sqrtf(r2) * 1.001f + 1e-6f * t->max_norm;
        }
        t->shadow_is_i8 = true;
        t->stats_valid = true;
        return PG_OK;
    }
    const size_t n_blk = (size_t)((t->rows + kPieceRows - 1) / kPieceRows) + 2;
    if (!t->d16) {
        const size_t bytes = (t->rows + 64) * (size_t)t->dim * 2;
        if (hipMalloc((void**)&t->d16, bytes) != hipSuccess) {
            (void)hipGetLastError();
            t->d16 = nullptr;
            t->shadow_failed = true;                  // stay on the exact scan
            return PG_OK;
        }
        PG_HIP(hipMemsetAsync(t->d16 + t->rows * (size_t)t->dim, 0, 64 * (size_t)t->dim * 2, ctx->stream));
    }
    if (!t->dnorm2 && hipMalloc((void**)&t->dnorm2, n_blk * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        t->dnorm2 = nullptr;
        t->shadow_failed = true;
        return PG_OK;
    }
    PG_HIP(hipMemsetAsync(t->dnorm2, 0, n_blk * sizeof(float), ctx->stream));
    PG_HIP(hipMemsetAsync(d_max, 0, 8, ctx->stream));
    if (t->dim == 64) table_shadow_kernel<64><<<grid, 256, 0, ctx->stream>>>(t->d, t->rows, t->d16, d_max, d_bad, t->dnorm2);
    else table_shadow_kernel<128><<<grid, 256, 0, ctx->stream>>>(t->d, t->rows, t->d16, d_max, d_bad, t->dnorm2);
    PG_HIP(hipGetLastError());
    PG_HIP(hipMemcpyAsync(ctx->h_status + 300, d_max, 8, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    float mx;
    memcpy(&mx, ctx->h_status + 300, 4);
    t->all_finite = ctx->h_status[301] == 0;
    t->max_norm = sqrtf(mx) * 1.0001f;
    t->shadow_is_i8 = false;
    t->stats_valid = true;
    return PG_OK;
}

// a job's status block, packed on the device by status_pack_kernel and copied to the host in ONE piece: [0] overflow flag,
// [1 + q] valid count of query q, then
constexpr uint32_t kPredStatsAt = 260;        // words [260, 270): the table's observation sums (five doubles)
constexpr uint32_t kI4mStatAt = 272;          // [272] the (row, query) pairs the 4-bit stage of a mid-batch pass let through,
                                              // [+1] suspects of the full pass's last launch, [+2] a hit-record area overflowed,
                                              // [+3] pairs that reached the exact re-scoring, [+4] candidates the pass collected
constexpr uint32_t kStatusCopyWords = kI4mStatAt + 5;
static_assert(kStatusCopyWords <= 300 && kStatusCopyWords <= kRecallStatusWords, "the status block ends in front of the words other calls keep in ctx->h_status");

// (launched <<<1, kMaxQueries>>>; null pointers = words that stay zero)
__global__ void status_pack_kernel(const uint32_t* __restrict__ overflow, uint32_t nq, uint32_t rec_ovf_word,
                                   const uint32_t* __restrict__ pred_stats, const uint32_t* __restrict__ pairs_word,
                                   const uint32_t* __restrict__ susp, const uint32_t* __restrict__ rescored,
                                   const uint32_t* __restrict__ cand_seen, uint32_t* __restrict__ out) {
    const uint32_t t = threadIdx.x;
    uint32_t v1 = (susp && t < nq) ? susp[t] : 0u, v3 = (rescored && t < nq) ? rescored[t] : 0u, v4 = (cand_seen && t < nq) ? cand_seen[t] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v1 += __shfl_xor(v1, off, 64);
        v3 += __shfl_xor(v3, off, 64);
        v4 += __shfl_xor(v4, off, 64);
    }
    __shared__ uint32_t s[3][4];
    if ((t & 63) == 0) { s[0][t >> 6] = v1; s[1][t >> 6] = v3; s[2][t >> 6] = v4; }
    out[1 + t] = t < nq ? overflow[1 + t] : 0u;
    if (t < 10) out[kPredStatsAt + t] = pred_stats ? pred_stats[t] : 0u;
    __syncthreads();
    if (t == 0) {
        out[0] = overflow[0];
        out[kI4mStatAt] = pairs_word ? *pairs_word : 0u;
        out[kI4mStatAt + 1] = s[0][0] + s[0][1] + s[0][2] + s[0][3];
        out[kI4mStatAt + 2] = overflow[rec_ovf_word];
        out[kI4mStatAt + 3] = s[1][0] + s[1][1] + s[1][2] + s[1][3];
        out[kI4mStatAt + 4] = s[2][0] + s[2][1] + s[2][2] + s[2][3];
    }
}

// the threshold model of a table (dim 128): mean and covariance of a row sample, built once per generation of the rows
static int ensure_pred_model(pg_ctx* ctx, const pg_table* tc) {
    pg_table* t = const_cast<pg_table*>(tc);
    std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
    if (t->pred_model) return PG_OK;
    const size_t bytes = (size_t)(128 + 128 * 128) * 4 + 5 * 8;   // mean | covariance (→ factor) | n, sum, sum2, min, total
    if (!t->d_pred && hipMalloc((void**)&t->d_pred, bytes) != hipSuccess) {
        (void)hipGetLastError();
        t->d_pred = nullptr;
        return PG_OK;                                 // no model: the pilot plan serves
    }
    PG_HIP(hipMemsetAsync(t->d_pred, 0, bytes, ctx->stream));
    const uint64_t n_sample = t->rows < (1u << 20) ? t->rows : (1u << 20);
    const uint64_t stride = t->rows / n_sample;
    pred_moments_kernel<<<(uint32_t)ctx->num_cus, 256, 0, ctx->stream>>>(t->d, t->rows, stride, n_sample, t->d_pred, t->d_pred + 128);
    pred_finish_kernel<<<128, 128, 0, ctx->stream>>>(t->d_pred, n_sample);
    pred_means_kernel<<<1, 128, 0, ctx->stream>>>(t->d_pred, n_sample);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(ctx->stream));        // other contexts of the device use the model from their own streams
    t->pred_model = true;
    t->pred_k = 0;
    t->pred_n = t->pred_sum = t->pred_sum2 = t->pred_total = 0.0;
    t->pred_min = 1e300;
    t->pred_backoff = 0;
    return PG_OK;
}

template <int DIM, int NQB, int WAVES, int SPLIT = 1, int VAR = 0, bool I8 = false, int QH = 1, int L2 = 0>
static int launch_screen(pg_ctx* ctx, const ScreenArgs& a) {
    int rc_attr;
    if ((rc_attr = ensure_dyn_lds(ctx, (const void*)screen_kernel<DIM, NQB, WAVES, SPLIT, VAR, I8, QH, L2>, kScreenLds))) return rc_attr;
    const uint32_t total = a.rb_end - a.rb_begin;
    uint32_t grid = (uint32_t)ctx->num_cus;
    const uint32_t need = (total * SPLIT + WAVES - 1) / WAVES;
    if (grid > need) grid = need;
#ifdef PG_SCREEN_PROFILE
    if (QH > 1) {
        static unsigned long long* dbg = nullptr;
        if (!dbg) hipMalloc(&dbg, 4096 * 64);
        ScreenArgs b = a;
        b.prof = dbg;
        screen_kernel<DIM, NQB, WAVES, SPLIT, VAR, I8, QH, L2><<<grid, 64 * WAVES, kScreenLds, ctx->stream>>>(b);
        static int calls = 0;
        if (total > 2000000 && ++calls == 12) {
            std::vector<unsigned long long> h(4096 * 8);
            hipMemcpy(h.data(), dbg, 4096 * 64, hipMemcpyDeviceToHost);
            for (int w : {0, 1, 4, 5, 8 * 100, 8 * 100 + 4, 8 * 255 + 3})
                fprintf(stderr, "[pg] screen wave %4d: wait %9llu lds+dma %9llu | half0 %9llu hit %9llu | half1 %9llu hit %9llu (cycles, %u blocks)\n", w,
                        h[w * 8], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5], (total + grid * WAVES - 1) / (grid * WAVES));
        }
        return PG_OK;
    }
#endif
    screen_kernel<DIM, NQB, WAVES, SPLIT, VAR, I8, QH, L2><<<grid, 64 * WAVES, kScreenLds, ctx->stream>>>(a);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

static int dispatch_screen(pg_ctx* ctx, uint32_t dim, bool i8, const ScreenArgs& a) {
    const bool wide = a.nq > 128;
    if (a.nx_rows) {                                 // squared-Euclidean recall, per-row test (rows of mixed norms)
        if (a.nq <= 32) return launch_screen<128, 1, 8, 1, 0, true, 1, 2>(ctx, a);
        if (a.nq <= 64) return launch_screen<128, 2, 8, 1, 0, true, 1, 2>(ctx, a);
        return launch_screen<128, 4, 8, 1, 0, true, 1, 2>(ctx, a);
    }
    if (a.blk_nxmin) {                               // squared-Euclidean recall: int8 shadow, <= 128 queries (recall_job_prepare)
        if (a.nq <= 32) return launch_screen<128, 1, 8, 1, 0, true, 1, 1>(ctx, a);
        if (a.nq <= 64) return launch_screen<128, 2, 8, 1, 0, true, 1, 1>(ctx, a);
        return launch_screen<128, 4, 8, 1, 0, true, 1, 1>(ctx, a);
    }
    if (i8) {                                        // int8 shadow (dim 128)
#ifdef PG_SCAN_VARIANTS
        const char* v = getenv("PG_SCREEN_VAR");     // developer ablation builds only
        if (wide && v && v[0] == '1') return launch_screen<128, 4, 8, 1, 1, true, 2>(ctx, a);
        if (wide && v && v[0] == '2') return launch_screen<128, 4, 8, 1, 2, true, 2>(ctx, a);
        if (wide && v && v[0] == '4') return launch_screen<128, 4, 8, 1, 4, true, 2>(ctx, a);
        if (wide && v && v[0] == '3') return launch_screen<128, 4, 8, 1, 3, true, 2>(ctx, a);     // no DMA (stale LDS): MFMA + tests
        if (wide && v && v[0] == '5') return launch_screen<128, 4, 8, 1, 5, true, 2>(ctx, a);     // no DMA, no tests: MFMA only
#endif
        if (wide) return launch_screen<128, 4, 8, 1, 0, true, 2>(ctx, a);
        if (a.nq <= 32) return launch_screen<128, 1, 8, 1, 0, true>(ctx, a);
        if (a.nq <= 64) return launch_screen<128, 2, 8, 1, 0, true>(ctx, a);
        return launch_screen<128, 4, 8, 1, 0, true>(ctx, a);
    }
#ifdef PG_SCAN_VARIANTS
    // PG_SCREEN_SPLIT=1 selects the wave-pair variant that fetches every block twice (measured slower: 8.3 vs
    // 6.9 ms per 256-query pass — DESIGN.md "what did not work"); developer ablation builds only.
    static const bool split = getenv("PG_SCREEN_SPLIT") != nullptr;
    if (wide && split) return dim == 64 ? launch_screen<64, 4, 8, 2>(ctx, a) : launch_screen<128, 4, 8, 2>(ctx, a);
#endif
    if (dim == 64) {
        if (wide) return launch_screen<64, 8, 4>(ctx, a);
        if (a.nq <= 32) return launch_screen<64, 1, 8>(ctx, a);
        if (a.nq <= 64) return launch_screen<64, 2, 8>(ctx, a);
        return launch_screen<64, 4, 8>(ctx, a);
    }
#ifdef PG_SCAN_VARIANTS
    {
        const char* v = getenv("PG_SCREEN_VAR");     // developer ablation builds only
        if (wide && v && v[0] == '1') return launch_screen<128, 8, 4, 1, 1>(ctx, a);
        if (wide && v && v[0] == '2') return launch_screen<128, 8, 4, 1, 2>(ctx, a);
        if (wide && v && v[0] == '3') return launch_screen<128, 8, 4, 1, 3>(ctx, a);
        if (wide && v && v[0] == '4') return launch_screen<128, 8, 4, 1, 4>(ctx, a);
        if (!wide && v && v[0] == '1') return launch_screen<128, 4, 8, 1, 1>(ctx, a);
        if (!wide && v && v[0] == '2') return launch_screen<128, 4, 8, 1, 2>(ctx, a);
        if (!wide && v && v[0] == '3') return launch_screen<128, 4, 8, 1, 3>(ctx, a);
        if (!wide && v && v[0] == '4') return launch_screen<128, 4, 8, 1, 4>(ctx, a);
    }
#endif
    if (wide) return launch_screen<128, 8, 4>(ctx, a);
    if (a.nq <= 32) return launch_screen<128, 1, 8>(ctx, a);
    if (a.nq <= 64) return launch_screen<128, 2, 8>(ctx, a);
    return launch_screen<128, 4, 8>(ctx, a);
}

// ---------------------------------------------------------------------------------------------
// One recall = one batch of queries against one table (one table pass when the first plan holds).
// Plans, tried in order until one completes without overflowing a candidate list:
//  pilot — the threshold comes from a jittered 1/S block sample (its K'-th best score, K' chosen six
//          sigma above the expected rank K/S, so it is below the true K-th best except with
//          negligible probability), then ONE full-rate pass over the whole table collects every row
//          at or above it.  Verified exactly: if a query ends with fewer than min(K, rows)
//          candidates the sample lied and the next plan runs.  ~1.5 K candidates per query instead
//          of K ln(rows/32768) for the growing-chunk plan.  On big tables the pass is split after its first
//          quarter and the thresholds are raised to the k2-th best candidate found so far (a 16x larger sample:
//          ~1.2 K candidates per query for the rest); the raised threshold is verified against the K-th best
//          score that comes out.  Batches of <= 4 queries stream the 4-bit shadow instead (recall_i4.hip).
//  grow  — geometric chunks, threshold refreshed after each (small tables, and the fallback).
//  safe  — bounded chunks that cannot overflow (adversarially ordered tables).
// A plan is ENQUEUED as a whole (no host synchronisation inside it) and VERIFIED afterwards from a few status
// words copied to pinned host memory: recall_dev_locked synchronises right away, the device-resident pipelines
// (pipeline.hip) defer the check to the end of the whole request batch, so the rank / fusion / sort stages of a
// batch are queued behind its recall without the GPU ever waiting for the host.
// ---------------------------------------------------------------------------------------------
enum RecallPlan { kPilot = 0, kGrow = 1, kSafe = 2, kPredict = 3 };

static inline uint64_t rs_cap_bound(uint32_t k) { return (uint64_t)k + kCandSlack; }

// rows that pass a recall's filter (one pass over the column: 0.4 GB per 100 M rows)
__global__ void filter_count_kernel(RowFilter f, uint64_t rows, unsigned long long* __restrict__ out) {
    unsigned long long n = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (uint64_t)gridDim.x * blockDim.x)
        n += row_filter_pass(f, (uint32_t)r) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}

// ---- selective filters: the admitted rows, in row order (stable compaction), gathered into a compact table ----
constexpr uint32_t kCompactBlockRows = 1024;
__device__ __forceinline__ bool filter_cmp(int op, long long v, long long val) {
    switch (op) {
        case 0: return v > val;
        case 1: return v >= val;
        case 2: return v < val;
        case 3: return v <= val;
        case 4: return v == val;
        default: return v != val;
    }
}
// bit i: row r0 + i passes (r0 a multiple of 4: one 16-B load of an int32 column, two of an int64 one)
__device__ __forceinline__ uint32_t row_filter_mask4(const RowFilter& f, uint64_t r0, uint64_t rows) {
    uint32_t m = 0;
    if (r0 + 4 <= rows) {
        if (f.dtype == PG_F_I64) {
            const longlong2 a = reinterpret_cast<const longlong2*>(reinterpret_cast<const long long*>(f.col) + r0)[0];
            const longlong2 b = reinterpret_cast<const longlong2*>(reinterpret_cast<const long long*>(f.col) + r0)[1];
            m = (filter_cmp(f.op, a.x, f.val) ? 1u : 0u) | (filter_cmp(f.op, a.y, f.val) ? 2u : 0u) | (filter_cmp(f.op, b.x, f.val) ? 4u : 0u) |
                (filter_cmp(f.op, b.y, f.val) ? 8u : 0u);
        } else {
            const int4 a = *reinterpret_cast<const int4*>(reinterpret_cast<const int32_t*>(f.col) + r0);
            m = (filter_cmp(f.op, a.x, f.val) ? 1u : 0u) | (filter_cmp(f.op, a.y, f.val) ? 2u : 0u) | (filter_cmp(f.op, a.z, f.val) ? 4u : 0u) |
                (filter_cmp(f.op, a.w, f.val) ? 8u : 0u);
        }
    } else {
        for (uint32_t i = 0; i < 4 && r0 + i < rows; ++i) m |= row_filter_pass(f, (uint32_t)(r0 + i)) ? 1u << i : 0u;
    }
    return m;
}
__global__ __launch_bounds__(256) void filter_block_count_kernel(RowFilter f, uint64_t rows, uint32_t* __restrict__ block_n) {
    __shared__ uint32_t ws[4];
    const uint64_t r0 = (uint64_t)blockIdx.x * kCompactBlockRows + threadIdx.x * 4;
    uint32_t n = r0 < rows ? (uint32_t)__popc(row_filter_mask4(f, r0, rows)) : 0u;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) block_n[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// The block counts' exclusive scan in two parallel levels: every group of 1024 counts is scanned in place by a workgroup of its
// own (its total → group_n[g]); one small workgroup then scans the <= 4096 group totals (grand total → group_n[ngroups]).  A
// block's offset is block_n[b] + group_n[b / 1024].
__device__ __forceinline__ uint32_t wg1024_exclusive_scan(uint32_t v, uint32_t* wsum, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t excl = incl - v, all = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < w) excl += wsum[j];
        all += wsum[j];
    }
    *total = all;
    return excl;
}
__global__ __launch_bounds__(1024) void filter_group_scan_kernel(uint32_t* __restrict__ block_n, uint32_t n, uint32_t* __restrict__ group_n) {
    __shared__ uint32_t wsum[16];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    const uint32_t v = i < n ? block_n[i] : 0u;
    uint32_t total;
    const uint32_t excl = wg1024_exclusive_scan(v, wsum, &total);
    if (i < n) block_n[i] = excl;
    if (threadIdx.x == 0) group_n[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void filter_total_scan_kernel(uint32_t* __restrict__ group_n, uint32_t ngroups) {
    __shared__ uint32_t wsum[16];
    uint32_t v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = threadIdx.x * 4 + j;
        v[j] = i < ngroups ? group_n[i] : 0u;
        s += v[j];
    }
    uint32_t total;
    uint32_t acc = wg1024_exclusive_scan(s, wsum, &total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = threadIdx.x * 4 + j;
        if (i < ngroups) group_n[i] = acc;
        acc += v[j];
    }
    if (threadIdx.x == 0) group_n[ngroups] = total;
}
// ids[block offset + rank inside the block] = row, ranks in row order: a thread takes 4 consecutive rows, a wave 256
__global__ __launch_bounds__(256) void filter_scatter_kernel(RowFilter f, uint64_t rows, const uint32_t* __restrict__ block_off,
                                                             const uint32_t* __restrict__ group_off, uint32_t* __restrict__ ids) {
    __shared__ uint32_t wn[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t r0 = (uint64_t)blockIdx.x * kCompactBlockRows + threadIdx.x * 4;
    const uint32_t m = r0 < rows ? row_filter_mask4(f, r0, rows) : 0u;
    const uint32_t c = (uint32_t)__popc(m);
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wn[w] = incl;
    __syncthreads();
    uint32_t pos = block_off[blockIdx.x] + group_off[blockIdx.x >> 10] + incl - c;
    for (int j = 0; j < w; ++j) pos += wn[j];
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i)
        if (m & (1u << i)) ids[pos++] = (uint32_t)(r0 + i);
}
// compact[i][:] = tab[ids[i]][:] (a wave per row-quad: 16 B per lane)
__global__ void compact_gather_kernel(const float* __restrict__ tab, const uint32_t* __restrict__ ids, uint32_t n, uint32_t dim,
                                      float* __restrict__ out) {
    const uint32_t q4 = dim / 4;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n * q4) return;
    const uint32_t r = (uint32_t)(i / q4), c = (uint32_t)(i % q4);
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(tab + (size_t)ids[r] * dim)[c];
}
__global__ void compact_gather_nx_kernel(const float* __restrict__ nx, const uint32_t* __restrict__ ids, uint32_t n, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n + 64) out[i] = i < n ? nx[ids[i]] : 0.0f;
}
__global__ void recall_pad_kernel(uint64_t* __restrict__ rows, float* __restrict__ scores, size_t n, bool l2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        rows[i] = ~0ull;
        scores[i] = l2 ? __builtin_inff() : -__builtin_inff();
    }
}
// local rows of the compact table → the table's global row ids (padding stays UINT64_MAX)
__global__ void compact_map_rows_kernel(uint64_t* __restrict__ rows, uint64_t n, const uint32_t* __restrict__ ids, uint64_t row_offset,
                                        uint64_t n_ids) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && rows[i] < n_ids) rows[i] = row_offset + ids[rows[i]];      // (padding, UINT64_MAX, stays)
}

// Count the rows `f` admits per 1024-row block (one pass over the column) and scan the counts — the scan is also the compaction's
// map (filter_scatter_kernel); buffers in scratch slot 12.  Synchronises the stream (the count comes back to the host).
int filter_count_locked(pg_ctx* ctx, const RowFilter& f, uint64_t rows, uint32_t** d_blk_out, uint32_t** d_grp_out, uint32_t* cblocks_out,
                        uint32_t* admitted_out) {
    const uint32_t cblocks = (uint32_t)((rows + kCompactBlockRows - 1) / kCompactBlockRows);
    const uint32_t cgroups = (cblocks + 1023) / 1024;
    if (cgroups > 4096) {
        set_error("filtered recall: table of %llu rows too large", (unsigned long long)rows);
        return PG_ERR_UNSUPPORTED;
    }
    void* cbuf;
    int rc;
    if ((rc = scratch_reserve(ctx, 12, ((size_t)cblocks + cgroups + 2) * 4, &cbuf))) return rc;
    uint32_t* d_blk = (uint32_t*)cbuf;
    uint32_t* d_grp = d_blk + cblocks;
    uint32_t admitted = 0;
    if (cblocks) {
        filter_block_count_kernel<<<cblocks, 256, 0, ctx->stream>>>(f, rows, d_blk);
        filter_group_scan_kernel<<<cgroups, 1024, 0, ctx->stream>>>(d_blk, cblocks, d_grp);
        filter_total_scan_kernel<<<1, 1024, 0, ctx->stream>>>(d_grp, cgroups);
        PG_HIP(hipGetLastError());
        PG_HIP(hipMemcpyAsync(&admitted, d_grp + cgroups, 4, hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(hipStreamSynchronize(ctx->stream));
    }
    *d_blk_out = d_blk;
    *d_grp_out = d_grp;
    *cblocks_out = cblocks;
    *admitted_out = admitted;
    return PG_OK;
}

// |x|^2 of every row, for the squared-Euclidean recall (lazily, cached until the next upload / fill; shared by the contexts
// of a device like the shadows)
int ensure_table_nx(pg_ctx* ctx, const pg_table* tc) {
    pg_table* t = const_cast<pg_table*>(tc);
    std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
    if (t->nx_valid) return PG_OK;
    const uint32_t nblocks = (uint32_t)((t->rows + kPieceRows - 1) / kPieceRows);
    if (!t->d_nx) PG_HIP(hipMalloc((void**)&t->d_nx, (t->rows + 64) * sizeof(float)));
    if (!t->d_nxmin) PG_HIP(hipMalloc((void**)&t->d_nxmin, ((size_t)nblocks + 64) * sizeof(float)));
    PG_HIP(hipMemsetAsync(t->d_nx + t->rows, 0, 64 * sizeof(float), ctx->stream));
    PG_HIP(hipMemsetAsync(t->d_nxmin + nblocks, 0, 64 * sizeof(float), ctx->stream));
    row_norm2_kernel<<<(uint32_t)((t->rows + 255) / 256), 256, 0, ctx->stream>>>(t->d, t->rows, t->dim, t->d_nx);
    void* p;
    int rc;
    if ((rc = scratch_reserve(ctx, 4, 4096, &p))) return rc;
    double* d_st = reinterpret_cast<double*>((char*)p + 2048);
    PG_HIP(hipMemsetAsync(d_st, 0, 16, ctx->stream));
    block_nxmin_kernel<<<(nblocks + 255) / 256, 256, 0, ctx->stream>>>(t->d_nx, t->rows, t->d_nxmin, nblocks, d_st);
    PG_HIP(hipGetLastError());
    double h_st[2] = {0.0, 0.0};
    PG_HIP(hipMemcpyAsync(h_st, d_st, 16, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    // What the per-block cutoff gives away: a row's cutoff sits (|x|^2 - min of its block) / 2 below its own, in inner-product
    // units whose spread over the table is about |x| |q| / sqrt(dim) ~ mean|x|^2 / sqrt(dim) for queries like the rows.  Normalised
    // rows: 0.  N(0, 1) rows: 1.5 spreads — then nearly every block holds suspects and the exact scan is the faster pass.
    const double mean_nx = t->rows ? h_st[1] / (double)t->rows : 0.0;
    t->l2_slack = mean_nx > 0.0 ? (float)((h_st[0] / (double)t->rows) * 0.5 / (mean_nx / sqrt((double)t->dim))) : 0.0f;
    if (ctx->knobs.debug_scan) fprintf(stderr, "[pg] squared-Euclidean recall: per-block cutoff slack %.3f score spreads\n", t->l2_slack);
    t->nx_valid = true;
    return PG_OK;
}

int recall_job_prepare(RecallJob* j) {
    pg_ctx* ctx = j->ctx;
    const pg_table* t = j->t;
    const Knobs& kn = ctx->knobs;
    int rc;
    if (j->l2) {
        if (t->dim != 64 && t->dim != 128) {
            set_error("recall (squared Euclidean): dim=%u unsupported (64 or 128)", t->dim);
            return PG_ERR_UNSUPPORTED;
        }
        if ((rc = ensure_table_nx(ctx, t))) return rc;
    }
    j->table_gen = t->generation.load(std::memory_order_relaxed);
    j->rows_qualified = (uint32_t)t->rows;
    if (j->filter.col && j->filter.admitted >= 0) {
        j->rows_qualified = (uint32_t)j->filter.admitted;
    } else if (j->filter.col) {
        // how many rows the filter admits: the plan check needs it (a filtered list may rightly be shorter than K), and the
        // pilot plan is only worth its launches when the sample holds enough of them
        void* p;
        if ((rc = scratch_reserve(ctx, 4, 4096, &p))) return rc;
        unsigned long long* d_n = reinterpret_cast<unsigned long long*>((char*)p + 3072);
        PG_HIP(hipMemsetAsync(d_n, 0, 8, ctx->stream));
        filter_count_kernel<<<(uint32_t)ctx->num_cus * 8, 256, 0, ctx->stream>>>(j->filter, t->rows, d_n);
        PG_HIP(hipGetLastError());
        unsigned long long h_n = 0;
        PG_HIP(hipMemcpyAsync(&h_n, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
        PG_HIP(hipStreamSynchronize(ctx->stream));
        j->rows_qualified = (uint32_t)h_n;
        j->filter.admitted = (long long)h_n;      // (the re-run of a failed query does not count again)
    }
    if ((rc = recall_scratch(ctx, t->dim, j->k, &j->rs))) return rc;
    j->d_count = j->rs.overflow + 1;              // status words in one block: one device → host copy
    j->rows = (uint32_t)t->rows;
    j->nblocks = (j->rows + kPieceRows - 1) / kPieceRows;
    j->scan_ms = j->total_ms = 0.0;
    j->scanned_rows = 0;
    j->scan_bytes = 0;
    j->scan_launches = 0;
    j->next_plan = 0;
    j->enqueued_plan = -1;
    j->failed.clear();

    // Policy: finite tables of dim <= 128 use the screened scan (int8 or bf16 filter + exact re-scoring) for every
    // batch size (knobs.screen_min = 0), HBM-bound up to 128 queries per pass; everything else rides the exact
    // fp32-MFMA scan in groups of <= 64 queries, one launch per group.
    bool screen = j->nq > kn.screen_min && t->dim <= 128 && !kn.recall_exact && !j->exact_only;
    if (screen) {
        if ((rc = ensure_table_stats(ctx, t))) return rc;
        if (!t->stats_valid || !t->all_finite) screen = false;      // no shadow (dim, memory) or non-finite rows
    }
    if (screen && t->screen_backoff) {                 // this table's screened plans kept overflowing: exact scan for now
        std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
        pg_table* tm = const_cast<pg_table*>(t);
        if (tm->screen_backoff) {
            tm->screen_backoff--;
            screen = false;
        }
    }
    // squared Euclidean: the int8 screen with per-block cutoffs for up to 128 queries (dim 128, int8 shadow); else the exact scan
    if (j->l2 && !(screen && t->dim == 128 && t->shadow_is_i8 && j->nq <= 128 && !kn.l2_exact)) screen = false;
    // rows of (nearly) one norm: one cutoff per 32-row block; beyond: the per-row test (three VALU instructions per bound instead of half a one)
    j->l2_per_row = j->l2 && screen && t->l2_slack > kn.l2_max_slack;
    j->screen = screen;
    j->screen4 = j->screen4m = false;
    j->n_plans = 0;
    j->stride = 1;
    j->sample_blocks = 0;
    j->k_pilot = 0;
    j->perm_mul = 1;
    const uint32_t rows = j->rows;
    const uint32_t full_blocks = rows / kPieceRows;           // the sample only uses whole blocks
    // 1/96 of the blocks: with the refinement step of the pilot plan (below) the sample's threshold only serves the
    // first quarter of the full pass, so a thinner, cheaper sample wins (256 requests: 291 vs 286 M items/s at 1/64,
    // 261 M at 1/32; before the refinement 1/64 was best)
    uint32_t want = full_blocks / 96;
    // >= 1M sample rows, but never more than an eighth of the table: small tables are launch-bound, and the pilot's
    // three scan launches beat the growing-chunk plan's nine (1M x 64, K = 200: 0.31 -> 0.13 ms per recall)
    const uint32_t floor_blocks = full_blocks / 8 < 32768 ? full_blocks / 8 : 32768;
    if (want < floor_blocks) want = floor_blocks;
    if (kn.pilot_fraction > 0.0) want = (uint32_t)(full_blocks * kn.pilot_fraction);
    j->stride = want ? full_blocks / want : 1;
    if (j->stride >= 2 && !kn.no_pilot) {
        j->sample_blocks = full_blocks / j->stride;
        const double m = (double)j->k * ((double)j->sample_blocks * kPieceRows / (double)rows);
        j->k_pilot = (uint32_t)ceil(m + ctx->knobs.pilot_sigmas * sqrt(m) + 8.0);
        // (a filter thins the sample by its selectivity: the sample must still hold 4 K' admitted rows)
        const double admitted = rows ? (double)j->rows_qualified / (double)rows : 1.0;
        if ((double)j->k_pilot * 4.0 <= (double)j->sample_blocks * kPieceRows * admitted) j->plans[j->n_plans++] = kPilot;
        j->perm_mul = 2654435761u % j->sample_blocks;          // golden-ratio step, made coprime below
        if (j->perm_mul < 2) j->perm_mul = 1;
        auto gcd = [](uint32_t x, uint32_t y) { while (y) { const uint32_t r = x % y; x = y; y = r; } return x; };
        while (gcd(j->perm_mul, j->sample_blocks) != 1) ++j->perm_mul;
    }
    j->plans[j->n_plans++] = kGrow;
    j->plans[j->n_plans++] = kSafe;
    if (j->skip_pilot && j->plans[0] == kPilot) j->next_plan = 1;    // the re-run of a query a sampled threshold failed
    // small batches: the pilot plan's full pass is HBM-bound on the shadow it streams — use the 4-bit one (recall_i4.hip)
    if (screen && t->dim == 128 && j->nq <= kI4MaxQueries && j->plans[0] == kPilot && !kn.no_screen_i4 &&
        rows >= kn.i4_min_rows && (uint64_t)kMaxQueries * rs_cap_bound(j->k) / kI4MaxQueries < 0xFFFFFFFFull) {
        if ((rc = ensure_table_i4(ctx, t))) return rc;
        // (measured at 100M x 128, int8 pass 2.1 ms: uniform rows, lambda 0.8: 1.37 / 1.38 / 1.59 / 1.66 ms at 1..4
        // queries; Gaussian rows, lambda 1.3: 1.46 / 1.59 / 1.78 / 2.02 — every query re-scores its own suspects)
        static const double kLamScale[kI4MaxQueries] = {1.0, 1.0, 0.88, 0.7};
        j->screen4 = t->i4_ok && (double)t->lam4 <= kn.i4_max_lambda * kLamScale[j->nq - 1];
    }
    // 1 .. 64 queries, inner product, int8 main shadow: the same 4-bit shadow through the matrix pipe, suspects thinned on the int8
    // shadow (recall_i4m.hip) — also for one or two queries (1.08 against the vector screen's 1.17 ms per 100 M rows: its int8 stage
    // re-scores a twentieth of the suspects)
    if (screen && !j->l2 && t->dim == 128 && t->shadow_is_i8 && j->nq >= kn.i4m_min_queries && j->nq <= kI4mMaxQueries &&
        j->nq <= kn.i4m_max_queries && j->plans[0] == kPilot && !kn.no_screen_i4m && rows >= kn.i4_min_rows) {
        if ((rc = ensure_table_i4(ctx, t))) return rc;
        // every pair the 4-bit stage lets through is one random 128-B read (~35 G/s in rescreen8_kernel): beyond i4m_max_pairs of them per pass
        // the int8 shadow's wider stream is the shorter pass.  The count per query is the table's own running average.
        // (two query blocks: the 4-bit stage is vector-ALU-bound, 1.5 instead of 1.2 ms per 100 M rows — the budget shrinks with the gain)
        j->screen4m = t->i4_ok && (double)t->lam4 <= kn.i4m_max_lambda &&
                      (double)t->i4m_pairs * j->nq <= kn.i4m_max_pairs * (j->nq > 32 ? 0.7 : 1.0);
        if (j->screen4m) j->screen4 = false;
    }
    // crowded rows (clustered embeddings: many suspects per answer): the int8 screen's suspects pass a two-digit refinement on the
    // residual shadow before the exact re-scoring (recall_r2.hip); passes with a 4-bit stage have their own int8 stage
    j->stage2 = false;
    if (screen && !j->l2 && t->dim == 128 && t->shadow_is_i8 && !j->screen4m && !j->screen4 && !kn.no_r2 &&
        (double)t->wide_susp > kn.r2_min_factor * (double)j->k) {
        if ((rc = ensure_table_r2(ctx, t))) return rc;
        j->stage2 = t->r2_ok;
    }
    // the threshold model: observe with every pilot-plan batch of a big int8-screened table; predict once the observed
    // quantile is tight (plan 0)
    j->predict = j->pred_observe = false;
    // (batches of <= 4 queries too: their full pass — the 4-bit shadow's — takes the same float thresholds, and the pilot's five
    //  launches are 0.15 ms of a lone request's 1.45)
    if (screen && !j->l2 && !j->filter.col && t->dim == 128 && t->shadow_is_i8 && j->plans[0] == kPilot && !j->skip_pilot && !kn.no_predict &&
        rows >= kn.predict_min_rows && j->k < rows / 64) {
        if ((rc = ensure_pred_model(ctx, t))) return rc;
        pg_table* tm = const_cast<pg_table*>(t);
        std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
        if (t->pred_model) {
            if (tm->pred_k != j->k) {                  // the observations belong to one K
                PG_HIP(hipMemsetAsync(t->d_pred + 128 + 128 * 128, 0, 24, ctx->stream));
                tm->pred_k = j->k;
                tm->pred_n = tm->pred_sum = tm->pred_sum2 = 0.0;
                tm->pred_backoff = 4;                  // (batches already in flight still report the old K's quantiles)
            }
            // (a mature model learns nothing from a lone request: its two launches and the 40-byte copy are spared)
            j->pred_observe = !(j->nq <= (uint32_t)kI4MaxQueries && t->pred_n >= 8192.0);
            if (tm->pred_backoff > 0) {
                --tm->pred_backoff;
            } else if (t->pred_n >= 1024.0) {
                const double mean = t->pred_sum / t->pred_n;
                const double var = t->pred_sum2 / t->pred_n - mean * mean;
                const double sd = var > 0.0 ? sqrt(var) : 0.0;
                // tight enough to beat the sample's own six-sigma margin (a threshold 3 % low in z doubles the rows that reach it)
                if (mean > 0.5 && sd <= 0.025 * mean) {
                    j->predict = true;
                    j->z_lo = mean - kn.predict_sigmas * sd - 0.002 * mean;
                    for (int i = j->n_plans; i > 0; --i) j->plans[i] = j->plans[i - 1];
                    j->plans[0] = kPredict;
                    j->n_plans++;
                }
            }
        }
    }
    return PG_OK;
}

namespace {
struct PlanRun {                     // the launches of one plan (helper of recall_job_enqueue)
    RecallJob* j;
    pg_ctx* ctx;
    const pg_table* t;
    RecallScratch& rs;
    uint32_t n_ev = 0;
    int cur = 0;
    bool susp2_clean = false;        // screen4m_prep_kernel left the mid-batch pass's second counters at zero (same)
    bool susp_clean = true;          // recall_init_kernel left the suspect counters at zero: the plan's first screened launch skips its memset
    bool no_i8 = false;              // the plan launches no int8 / bf16 screen (thresholds predicted, full pass on the 4-bit shadow):
                                     // its integer-unit thresholds are not needed
    bool last_full_screened = false; // the plan's last scan launch was a screened one ...
    bool last_was_i4m = false;       // ... of the mid-batch kind (its survivors are counted in susp2_cnt)
    bool last_was_r2 = false;        // ... followed by the refinement stage (its survivors are counted in susp2w_cnt)
    bool exact_chunks = false;       // every chunk on the exact scan: the safe plan of a FILTERED recall.  Its thresholds stay open until K
                                     // admitted rows were seen — under a selective filter, for many chunks — and a screened chunk with open
                                     // thresholds makes every row a suspect of every query: the hit-record regions are sized for the bounded
                                     // chunk spread evenly over the waves, the waves' shares are not even.  The exact scan stages admitted
                                     // rows only, at most the chunk's rows per query.

    explicit PlanRun(RecallJob* job) : j(job), ctx(job->ctx), t(job->t), rs(job->rs) {}

    // one scan launch over logical blocks [rb, rb+cb) of a stride-`st` view, bracketed by HIP events:
    // their sum is the per-pass duration of the dominant kernel that bench.py prices against HBM
    int scan_range(uint32_t rb, uint32_t cb, uint32_t st, bool thr_is_open, bool allow_i4 = false) {
        std::vector<hipEvent_t>& pool = *j->events;
        while (pool.size() < 2 * (size_t)(n_ev + 1)) {
            hipEvent_t e;
            PG_HIP(hipEventCreate(&e));
            pool.push_back(e);
        }
        if (j->timers) PG_HIP(hipEventRecord(pool[2 * n_ev], ctx->stream));
        const uint32_t nq = j->nq;
        if (j->screen && !thr_is_open && !exact_chunks) {
            ScreenArgs sa;
            sa.tab16 = t->shadow_is_i8 ? (const void*)t->d8 : (const void*)t->d16;
            sa.blk_norm2 = t->dnorm2;
            sa.eps_unit = rs.eps;
            sa.qb16 = rs.qb16;
            sa.thr_screen = rs.thr_screen;
            sa.susp_cnt = rs.susp_cnt;
            sa.susp = rs.susp;
            sa.overflow = rs.overflow;
            sa.cap = rs.cap;
            sa.nq = nq;
            sa.rb_begin = rb;
            sa.rb_end = rb + cb;
            sa.row_end = j->rows;
            sa.stride = st;
            sa.perm_mul = j->perm_mul;
            sa.perm_mod = j->sample_blocks;
            int rc2;
            // > 128 queries on the int8 shadow: the scan leaves hit records, decoded into the suspect lists below
            const bool records = t->shadow_is_i8 && nq > 128;
            sa.rec = nullptr;
            sa.rec_cnt = nullptr;
            sa.rec_cap = 0;
            sa.rec_pool = nullptr;
            sa.rec_pool_cap = sa.rec_waves = 0;
            sa.early_share = records ? ctx->knobs.screen_early_share : ctx->knobs.screen_early_share_narrow;
            sa.blk_nxmin = j->l2 && !j->l2_per_row ? t->d_nxmin : nullptr;
            sa.nx_rows = j->l2 && j->l2_per_row ? t->d_nx : nullptr;
            sa.l2_b = j->l2 ? rs.thr_ref : nullptr;    // (no refinement step under this metric: its buffer carries B_q)
            const uint32_t rec_waves = (uint32_t)ctx->num_cus * 8u;
            if (records) {
                // a region per wave: four times the share of 256 x K suspects a wave expects, never below the records the safe
                // plan's bounded chunks can leave in one region
                const uint64_t rsc = t->rec_scale ? t->rec_scale : 1;     // (tables whose batches overflowed these areas got larger ones)
                uint32_t cap_w = (uint32_t)(((uint64_t)kMaxQueries * j->k * 4 * rsc / rec_waves + 255) / 256 * 256);
                // (the safe plan's chunk — kCandSlack / 2 rows — with EVERY row a suspect of every query, e.g. a table in ascending
                //  score order: 8 x 64 records per block, and the blocks of a SIMD's pair of waves split as screen_kernel splits
                //  them — the larger share decides.  Sized for an even split, the early wave's fifth block went to the spill pool,
                //  which at a small K is small: "overflow in safe mode" on such a table at K = 10, 256 queries.)
                const uint32_t pairs = (uint32_t)ctx->num_cus * 4u;
                const uint32_t pb = (kCandSlack / 2 / kPieceRows + pairs - 1) / pairs;
                const uint32_t es = sa.early_share > 512 ? sa.early_share : 1024 - sa.early_share;
                uint32_t eb = (uint32_t)(((uint64_t)pb * es + 512) >> 10) + 1;
                if (eb > pb) eb = pb;
                const uint32_t floor_w = eb * 512;
                if (cap_w < floor_w) cap_w = floor_w;
                // + a spill pool for tables whose best rows sit together: room for all of 256 x 2 K suspects
                const uint32_t pool_cap = (uint32_t)(((uint64_t)kMaxQueries * j->k * 2 * rsc + kRecPoolSlice - 1) / kRecPoolSlice * kRecPoolSlice);
                const size_t head = ((size_t)(rec_waves + 1) * 4 + 255) & ~(size_t)255;
                void* p;
                if ((rc2 = scratch_reserve(ctx, 11, head + ((size_t)rec_waves * cap_w + pool_cap) * kRecBytes, &p))) return rc2;
                sa.rec_cnt = (uint32_t*)p;
                sa.rec = (char*)p + head;
                sa.rec_cap = cap_w;
                sa.rec_pool = sa.rec + (size_t)rec_waves * cap_w * kRecBytes;
                sa.rec_pool_cap = pool_cap;
                sa.rec_waves = rec_waves;
                PG_HIP(hipMemsetAsync(sa.rec_cnt, 0, (size_t)(rec_waves + 1) * 4, ctx->stream));
            }
            if (!susp_clean) PG_HIP(hipMemsetAsync(rs.susp_cnt, 0, sizeof(uint32_t) * kMaxQueries, ctx->stream));
            susp_clean = false;
            last_full_screened = true;
            last_was_r2 = false;
            // the full pass of a small batch streams the 4-bit shadow; its few suspect lists share the whole buffer
            // (whole 64-row groups: a range starts on an even block and ends on one or at the table's end)
            const bool i4 = j->screen4 && allow_i4 && st == 1 && (rb & 1) == 0 && (((rb + cb) & 1) == 0 || rb + cb == j->nblocks);
            const bool i4m = j->screen4m && allow_i4 && st == 1 && (rb & 1) == 0 && (((rb + cb) & 1) == 0 || rb + cb == j->nblocks);
            const uint32_t scap = i4 ? rs.cap * (uint32_t)(kMaxQueries / kI4MaxQueries) : rs.cap;
            const uint64_t r_begin = (uint64_t)rb * kPieceRows;
            const uint64_t r_end = (uint64_t)(rb + cb) * kPieceRows < j->rows ? (uint64_t)(rb + cb) * kPieceRows : j->rows;
            if (i4) {
                if ((rc2 = screen4_launch(ctx, t, rs, nq, (uint32_t)r_begin, (uint32_t)r_end, scap, j->l2 ? rs.pred_ms : nullptr))) return rc2;
                j->scan_bytes += (r_end - r_begin) * (j->l2 ? 72 : 68);      // (squared Euclidean: + the row's |x|^2)
            } else if (i4m) {
                // stage-1 suspects share the whole buffer ([nq][4 cap]); what passes the int8 stage lands in susp2 ([nq][cap])
                if ((rc2 = screen4m_launch(ctx, t, rs, nq, (uint32_t)r_begin, (uint32_t)r_end, rs.cap * (uint32_t)(kMaxQueries / kI4mMaxQueries), susp2_clean))) return rc2;
                susp2_clean = false;
                j->scan_bytes += (r_end - r_begin) * 68;
                last_was_i4m = true;
            } else {
                last_was_i4m = false;
                if ((rc2 = dispatch_screen(ctx, t->dim, t->shadow_is_i8, sa))) return rc2;
                if (records) {
                    screen_decode_kernel<<<rec_waves / 8 + sa.rec_pool_cap / kRecPoolSlice, 1024, 0, ctx->stream>>>(
                        sa.rec, sa.rec_cnt, sa.rec_cap, rec_waves, sa.rec_pool, sa.rec_pool_cap, rs.thr_screen, j->rows,
                                                                            rs.susp_cnt, rs.susp, rs.cap, rs.overflow);
                    PG_HIP(hipGetLastError());
                }
                j->scan_bytes += (uint64_t)cb * kPieceRows * t->dim * (t->shadow_is_i8 ? 1 : 2);
            }
            j->scanned_rows += (uint64_t)cb * kPieceRows;
            // exact re-scoring of the launch's suspects → candidate keys (grid.x strides over each list)
            const dim3 rg(i4 ? screen4_rescore_blocks() : kRescoreBlocksPerQuery, nq);
            const bool r2 = j->stage2 && t->shadow_is_i8 && !i4 && !i4m && !j->l2 && t->dim == 128;
            if (r2) {
                // crowded rows: the suspects once more with two more digits (recall_r2.hip); what is left goes to the exact re-scoring
                if ((rc2 = rescreen16_launch(ctx, t, rs, nq))) return rc2;
                last_was_r2 = true;
                rescore_kernel<128><<<rg, 256, 0, ctx->stream>>>(t->d, rs.qpad, rs.thr, rs.susp2, rs.susp2w_cnt, rs.cap,
                                                                 rs.cnt, rs.cand[cur], rs.overflow, rs.cap, j->rows, nullptr, nullptr, j->filter);
            } else if (j->l2)
                rescore_kernel<128, true><<<rg, 256, 0, ctx->stream>>>(t->d, rs.qpad, rs.thr, rs.susp, rs.susp_cnt, rs.cap, rs.cnt,
                                                                       rs.cand[cur], rs.overflow, scap, j->rows, t->d_nx, rs.pred_ms, j->filter);
            else if (t->dim == 64)
                rescore_kernel<64><<<rg, 256, 0, ctx->stream>>>(t->d, rs.qpad, rs.thr, rs.susp, rs.susp_cnt, rs.cap,
                                                                rs.cnt, rs.cand[cur], rs.overflow, scap, j->rows, nullptr, nullptr, j->filter);
            else if (i4m)
                rescore_kernel<128><<<rg, 256, 0, ctx->stream>>>(t->d, rs.qpad, rs.thr, rs.susp2, rs.susp2_cnt, rs.cap,
                                                                 rs.cnt, rs.cand[cur], rs.overflow, rs.cap, j->rows, nullptr, nullptr, j->filter);
            else
                rescore_kernel<128><<<rg, 256, 0, ctx->stream>>>(t->d, rs.qpad, rs.thr, rs.susp, rs.susp_cnt, rs.cap,
                                                                 rs.cnt, rs.cand[cur], rs.overflow, scap, j->rows, nullptr, nullptr, j->filter);
            PG_HIP(hipGetLastError());
        } else {
            last_full_screened = false;
            // exact scan (the first chunk of a screened recall too: its threshold is still -inf,
            // every row is a candidate and there is nothing to screen) in groups of <= 64 queries,
            // one launch; blockIdx.y walks the groups
            ScanArgs a;
            a.tab = t->d;
            a.qpad = rs.qpad;
            a.thr = rs.thr;
            a.cnt = rs.cnt;
            a.cand = rs.cand[cur];
            a.overflow = rs.overflow;
            a.cap = rs.cap;
            a.nq = nq;
            a.group_q = nq > (uint32_t)kMaxQueriesExact ? (uint32_t)kMaxQueriesExact : 0u;
            a.nq_launch = a.group_q ? a.group_q : nq;
            a.rb_begin = rb;
            a.rb_end = rb + cb;
            a.row_end = j->rows;
            a.stride = st;
            a.perm_mul = j->perm_mul;
            a.perm_mod = j->sample_blocks;
            a.nx = j->l2 ? t->d_nx : nullptr;
            a.nqv = j->l2 ? rs.pred_ms : nullptr;
            a.filter = j->filter;
            int rc2;
            if ((rc2 = dispatch_scan(ctx, t->dim, a))) return rc2;
            j->scan_bytes += (uint64_t)cb * kPieceRows * t->dim * 4;
            j->scanned_rows += (uint64_t)cb * kPieceRows;
        }
        if (j->timers) PG_HIP(hipEventRecord(pool[2 * n_ev + 1], ctx->stream));
        ++n_ev;
        return PG_OK;
    }
    // keep the best `kk` candidates per query, refresh the thresholds, swap the ping-pong lists
    // last: no launch screens against these thresholds any more (the screen's integer cutoffs are not derived)
    int refresh(uint32_t kk, bool last = false) {
        int rc2;
        if ((rc2 = launch_select(ctx, j->nq, rs.cand[cur], rs.cand[cur ^ 1], rs.cnt, rs.thr, rs.cap, kk, 0, rs.cnt_seen))) return rc2;
        if (j->screen && !no_i8 && !(last && !j->l2)) {
            if (j->l2) screen_thr8_l2_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.thr, rs.eps, rs.qscale, rs.pred_ms, t->s8, t->max_norm,
                                                                                rs.thr_screen, rs.thr_ref, j->l2_per_row ? 1 : 0);
            else if (t->shadow_is_i8) screen_thr8_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.thr, rs.eps, rs.qscale, t->s8, rs.thr_screen);
            else screen_thr_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.thr, rs.eps, t->max_norm, rs.thr_screen);
            PG_HIP(hipGetLastError());
        }
        cur ^= 1;
        return PG_OK;
    }
    // raise the thresholds to every query's kk-th best candidate so far (lists untouched)
    int refine(uint32_t kk) {
        int rc2;
        if ((rc2 = launch_select(ctx, j->nq, rs.cand[cur], rs.cand[cur ^ 1], rs.cnt, rs.thr, rs.cap, kk, 1))) return rc2;
        if (j->screen) {
            if (t->shadow_is_i8) screen_thr8_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.thr, rs.eps, rs.qscale, t->s8, rs.thr_screen);
            else screen_thr_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.thr, rs.eps, t->max_norm, rs.thr_screen);
            PG_HIP(hipGetLastError());
        }
        return PG_OK;
    }
    // geometric chunks over `nb` logical blocks of a stride-`st` view, keeping the best `kk`
    int grow_scan(uint32_t nb, uint32_t st, uint32_t kk, double growth, bool safe) {
        uint32_t rb = 0;
        while (rb < nb) {
            uint32_t chunk_rows;
            if (rb == 0) {
                // every row of the first chunk becomes a candidate of every query: keep it just large
                // enough to seed a meaningful threshold (8 x kk rows, 2048..32768)
                chunk_rows = next_pow2(8 * kk);
                if (chunk_rows < 2048) chunk_rows = 2048;
                if (chunk_rows > kFirstChunkRows || chunk_rows < kk) chunk_rows = kFirstChunkRows;
            } else {
                // chunk = (growth-1) x rows seen: every chunk stages ≈ (growth-1)*kk candidates per query
                const uint64_t grow = (uint64_t)((double)rb * kPieceRows * (growth - 1.0));
                chunk_rows = grow > 0xFFFFFFE0ull ? 0xFFFFFFE0u : (uint32_t)grow;
            }
            if (safe && chunk_rows > kCandSlack / 2) chunk_rows = kCandSlack / 2;
            uint32_t cb = chunk_rows / kPieceRows;
            if (cb > nb - rb) cb = nb - rb;
            int rc2;
            if ((rc2 = scan_range(rb, cb, st, rb == 0))) return rc2;
            rb += cb;
            if ((rc2 = refresh(kk))) return rc2;
        }
        return PG_OK;
    }
};
}  // namespace

int recall_job_enqueue(RecallJob* j) {
    pg_ctx* ctx = j->ctx;
    const pg_table* t = j->t;
    const Knobs& kn = ctx->knobs;
    if (j->next_plan >= j->n_plans) {
        set_error("recall: candidate overflow in safe mode (internal error)");
        return PG_ERR_DEVICE;
    }
    const int plan = j->plans[j->next_plan];
    int rc;
    // another call on this context may have grown the scratch arenas since recall_job_prepare (a re-plan reaches
    // here long after it): fetch the pointers again rather than trust the cached ones
    if ((rc = recall_scratch(ctx, t->dim, j->k, &j->rs))) return rc;
    j->d_count = j->rs.overflow + 1;
    PlanRun r(j);
    RecallScratch& rs = j->rs;
    recall_init_kernel<<<(kMaxQueries * t->dim + 255) / 256, 256, 0, ctx->stream>>>(
        j->d_queries, j->nq, t->dim, rs.qpad, rs.thr, rs.cnt, rs.overflow);
    PG_HIP(hipGetLastError());
    if (j->l2) {
        query_norm2_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.qpad, t->dim, rs.pred_ms);      // (the threshold model is off: its buffer is free)
        PG_HIP(hipGetLastError());
    }
    // thresholds predicted and the full pass on the 4-bit shadow: the plan launches no int8 screen, so neither the int8 query
    // fragments nor the integer-unit thresholds are needed (two launches less in front of a lone request, two behind)
    r.no_i8 = plan == kPredict && j->screen4;
    if (j->screen) {
        if (!r.no_i8) {
            if (t->shadow_is_i8)
                screen_prep8_kernel<<<j->nq <= 32 ? 1u : (j->nq <= 64 ? 2u : (j->nq <= 128 ? 4u : 8u)), 128, 0, ctx->stream>>>(
                    rs.qpad, t->dim, t->max_norm, t->resid8, rs.qb16, rs.eps, rs.qscale);     // grid = the scan's NQB x QH
            else
                screen_prep_kernel<<<(kScreenMaxNQB * (t->dim / 16) * 64 + 255) / 256, 256, 0, ctx->stream>>>(
                    rs.qpad, t->dim, rs.qb16, rs.eps);
            PG_HIP(hipGetLastError());
        }
        if (j->screen4 && (rc = screen4_prep_launch(ctx, t, rs))) return rc;
        if (j->screen4m) {
            if ((rc = screen4m_prep_launch(ctx, rs, j->nq))) return rc;
            r.susp2_clean = true;
        }
        if (j->stage2 && (rc = rescreen16_prep_launch(ctx, t, rs, j->nq))) return rc;
    }
    const bool observe = j->pred_observe && (plan == kPilot || plan == kPredict);
    if (observe || plan == kPredict) {
        // mean and sigma of every query's scores; under prediction the same launch writes the first thresholds
        pred_query_kernel<<<j->nq, 128, 0, ctx->stream>>>(rs.qpad, t->d_pred, rs.pred_ms, plan == kPredict ? rs.thr : nullptr,
                                                          (float)j->z_lo);
        PG_HIP(hipGetLastError());
    }
    while (j->events->size() < 2) {
        hipEvent_t e;
        PG_HIP(hipEventCreate(&e));
        j->events->push_back(e);
    }
    bool refined = false;
    // events 0/1 of the pool bracket the whole plan; PlanRun's launches use the pairs after them
    r.n_ev = 1;
    j->timers = !ctx->timers_off;
    if (j->timers) PG_HIP(hipEventRecord((*j->events)[0], ctx->stream));
    if (plan == kPredict && !r.no_i8) {
        // no sample: the first thresholds are the model's (pred_query_kernel above), here in the int8 screen's units
        screen_thr8_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.thr, rs.eps, rs.qscale, t->s8, rs.thr_screen);
        PG_HIP(hipGetLastError());
    }
    if (plan == kPilot || plan == kPredict) {
      if (plan == kPilot) {
        // The sample itself is streamed in two launches when it is screened: a seed of `seed_rows` rows
        // (exact, every row a candidate) whose m0-th best score becomes the threshold of ONE screened
        // launch over the rest of the sample; m0 is chosen so that fewer than K' sample rows reaching
        // that threshold has probability < 1e-9 (Poisson tail of seed hits among the sample's top K');
        // should it happen anyway, select_kernel leaves the threshold at -inf, the full pass overflows
        // and the next plan takes over.  (Measured against geometric chunks over the sample: 1.0 → 0.7 ms.)
        const uint32_t seed_rows = kn.seed_rows;
        const uint64_t sample_rows = (uint64_t)j->sample_blocks * kPieceRows;
        if (j->screen && kn.pilot_growth <= 0.0 && sample_rows > 8ull * seed_rows && j->k_pilot < seed_rows / 4) {
            const double mu = (double)seed_rows * (double)j->k_pilot / (double)sample_rows;
            uint32_t m0 = 1;
            for (double term = exp(-mu), cdf = term; 1.0 - cdf > 1e-9 && m0 < seed_rows; ++m0) {
                term *= mu / (double)m0;          // P(X = m0)
                cdf += term;                      // P(X <= m0)  →  loop ends with P(X >= m0+1) <= 1e-9
            }
            ++m0;
            const uint32_t sb = seed_rows / kPieceRows;
            if ((rc = r.scan_range(0, sb, j->stride, true))) return rc;
            if ((rc = r.refresh(m0))) return rc;
            if ((rc = r.scan_range(sb, j->sample_blocks - sb, j->stride, false))) return rc;
            if ((rc = r.refresh(j->k_pilot))) return rc;
        } else {
            // geometric chunks over the sample (exact scan, or pilot_growth set: A/B runs)
            if ((rc = r.grow_scan(j->sample_blocks, j->stride, j->k_pilot, kn.pilot_growth > 0.0 ? kn.pilot_growth : 8.0, false))) return rc;
        }
      }
        if (plan == kPilot) PG_HIP(hipMemsetAsync(rs.cnt, 0, sizeof(uint32_t) * kMaxQueries, ctx->stream));    // (kPredict: still zero from recall_init_kernel)
        // The sample's threshold is deliberately low (K' = m + 6 sqrt(m) + 8 of a 1/64 sample: ~1.8 K rows reach it where K
        // are needed), and every row that reaches it costs a suspect's hit path and an exact re-scoring — a quarter of
        // the 256-query pass.  After the first quarter of the table the candidates found so far ARE a 16x larger sample:
        // their k2-th best (k2 from the same formula) is a much tighter threshold for the other three quarters.  A
        // contiguous prefix is not a random sample: if the prefix holds more than its share of the best rows the raised
        // threshold can exceed the true K-th score — which is CHECKED after the pass (refine_verify_kernel: the K-th
        // best score that came out must reach the raised threshold), and such a query is re-run like one whose sample
        // threshold was too high.  Randomly ordered rows fail with the sample's own probability (six sigma).
        const uint32_t nb_q = (j->nblocks / 4) & ~1u;      // (even: the 4-bit screens walk whole 64-row pieces)
        const double m2 = (double)j->k * 0.25;
        const uint32_t k2 = (uint32_t)ceil(m2 + kn.pilot_sigmas * sqrt(m2) + 8.0);
        // (not behind a predicted threshold: that one already sits tighter than what a quarter of the table can certify —
        //  rank ~1.1 K against k2's ~1.18 K — so the split would only add a launch boundary)
        const bool refine = j->screen && !j->l2 && !j->screen4 && !kn.no_refine && j->rows >= kn.refine_min_rows && nb_q >= 64 &&
                            k2 < j->k && t->prefix_failures < 2 && plan != kPredict;
        if (refine) {
            if ((rc = r.scan_range(0, nb_q, 1, false, true))) return rc;
            if ((rc = r.refine(k2))) return rc;
            PG_HIP(hipMemcpyAsync(rs.thr_ref, rs.thr, sizeof(float) * kMaxQueries, hipMemcpyDeviceToDevice, ctx->stream));
            refined = true;
            if ((rc = r.scan_range(nb_q, j->nblocks - nb_q, 1, false, true))) return rc;
        } else if ((rc = r.scan_range(0, j->nblocks, 1, false, true))) {
            return rc;
        }
        // (statistics: the candidates the full pass collected, before the select keeps K of each list — a predicted threshold that
        //  admits many times K is no use to the table, recall_job_check)
        // (select_kernel notes them in rs.cnt_seen)
        if ((rc = r.refresh(j->k, true))) return rc;
    } else {
        // measured: growth 4 is best for the exact scan, 2 for the screened scan whose re-scoring
        // gathers 512 B per staged candidate
        r.exact_chunks = plan == kSafe && j->filter.col;
        const double growth = kn.chunk_growth > 1.0 ? kn.chunk_growth : (j->screen && !r.exact_chunks ? 2.0 : 4.0);
        if ((rc = r.grow_scan(j->nblocks, 1, j->k, growth, plan == kSafe))) return rc;
    }
    if (j->timers) PG_HIP(hipEventRecord((*j->events)[1], ctx->stream));
    if ((rc = final_launch(ctx, rs.cand[r.cur], rs.cnt, rs.cap, j->nq, j->k, t->row_offset, j->d_out_rows,
                           j->d_out_scores, j->d_count)))
        return rc;
    if (t->d_row_map) {                                // a filtered view answers with its source's row ids
        const uint64_t n = (uint64_t)j->nq * j->k;
        compact_map_rows_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, ctx->stream>>>(j->d_out_rows, n, t->d_row_map, t->map_offset, t->rows);
        PG_HIP(hipGetLastError());
    }
    if (j->l2) {                                       // the lists were ranked by -d: the distances go out
        const uint64_t n = (uint64_t)j->nq * j->k;
        negate_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, ctx->stream>>>(j->d_out_scores, n);
        PG_HIP(hipGetLastError());
    }
    if (refined) {
        // Two thresholds were in force during the full pass, so "K candidates were found" no longer proves that none
        // of the true top K was rejected: the raised one must not exceed the K-th best score that came out — then at
        // least K rows reach it, and every row it rejected is below K rows that were kept.  A query that fails reports
        // zero items, which the plan check reads as "re-run this query" (the same path as a sample threshold that
        // turned out too high).
        refine_verify_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.thr, rs.thr_ref, j->d_count, j->nq);
        PG_HIP(hipGetLastError());
    }
    if (observe) {
        // the K-th best scores that came out (rs.thr after the last refresh) as quantiles of the model; queries that
        // failed report 0 items and do not count.  The sums travel to the host with the status words.
        double* stats = reinterpret_cast<double*>(t->d_pred + 128 + 128 * 128);
        pred_update_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(rs.thr, rs.pred_ms, j->d_count, j->nq, j->k < j->rows ? j->k : j->rows, stats);
        PG_HIP(hipGetLastError());
    }
    j->susp_stat = j->screen && (plan == kPilot || plan == kPredict) && r.last_full_screened;
    j->stat_wide = j->susp_stat && !r.last_was_i4m;
    {
        // the status words in one block, one copy (they were six copies and two sums of their own: ~40 us of a small batch's step):
        // [kI4mStatAt] the pairs a mid-batch pass's 4-bit stage let through (zero otherwise), [+ 1] the suspects its int8 stage / the
        // int8 screen handed on in the full pass's last launch, [+ 3] what reached the exact re-scoring: the same, or the survivors
        // of the refinement stage, [+ 4] the candidates the pass collected (select_kernel notes them before it keeps K)
        const uint32_t* const susp = j->susp_stat ? (r.last_was_i4m ? rs.susp2_cnt : rs.susp_cnt) : nullptr;
        status_pack_kernel<<<1, kMaxQueries, 0, ctx->stream>>>(
            rs.overflow, j->nq, kRecOvfWord, observe ? reinterpret_cast<const uint32_t*>(t->d_pred + 128 + 128 * 128) : nullptr,
            j->susp_stat && r.last_was_i4m ? rs.susp2_cnt + kI4mMaxQueries : nullptr, susp, j->susp_stat && r.last_was_r2 ? rs.susp2w_cnt : susp,
            j->susp_stat ? rs.cnt_seen : nullptr, rs.status);
        PG_HIP(hipGetLastError());
    }
    if (j->d_out_count)
        PG_HIP(hipMemcpyAsync(j->d_out_count, j->d_count, 4 * j->nq, hipMemcpyDeviceToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(j->h_status, rs.status, 4 * (size_t)kStatusCopyWords, hipMemcpyDeviceToHost, ctx->stream));
    j->n_ev = r.n_ev;
    j->refined = refined;
    j->enqueued_plan = plan;
    j->observed = observe;
    j->next_plan++;
    return PG_OK;
}

int recall_job_check(RecallJob* j, bool* ok_out) {
    pg_ctx* ctx = j->ctx;
    const int plan = j->enqueued_plan;
    float ms = 0.f;
    if (j->timers) PG_HIP(hipEventElapsedTime(&ms, (*j->events)[0], (*j->events)[1]));
    j->total_ms += ms;
    for (uint32_t i = 1; j->timers && i < j->n_ev; ++i) {
        PG_HIP(hipEventElapsedTime(&ms, (*j->events)[2 * i], (*j->events)[2 * i + 1]));
        j->scan_ms += ms;
        if (ctx->knobs.debug_scan) fprintf(stderr, "[pg] plan %d scan launch %u: %.3f ms\n", plan, i - 1, ms);
    }
    if (ctx->knobs.debug_scan && j->screen) {          // suspects of the last screened launch vs K (developer aid)
        uint32_t sc[4] = {0, 0, 0, 0};
        PG_HIP(hipMemcpy(sc, j->rs.susp_cnt, sizeof sc, hipMemcpyDeviceToHost));
        fprintf(stderr, "[pg] plan %d last screened launch: suspects of queries 0-3: %u %u %u %u (K = %u)\n", plan, sc[0], sc[1], sc[2], sc[3], j->k);
        if (j->t->shadow_is_i8 && j->nq > 128 && ctx->scratch[11].p) {     // hit records: the fullest wave region, the spill pool
            const uint32_t waves = (uint32_t)ctx->num_cus * 8u;
            std::vector<uint32_t> rc(waves + 1);
            PG_HIP(hipMemcpy(rc.data(), ctx->scratch[11].p, rc.size() * 4, hipMemcpyDeviceToHost));
            uint32_t mx = 0;
            uint64_t sum = 0;
            for (uint32_t w = 0; w < waves; ++w) { mx = rc[w] > mx ? rc[w] : mx; sum += rc[w]; }
            fprintf(stderr, "[pg] plan %d last screened launch: hit records %llu in wave regions (fullest %u), %u in the spill pool\n", plan,
                    (unsigned long long)sum, mx, rc[waves]);
        }
    }
    j->scan_launches += j->n_ev - 1;
    bool ok = j->h_status[0] == 0;
    j->failed.clear();
    if (ok && (plan == kPilot || plan == kPredict)) {
        const uint32_t have = j->filter.col ? j->rows_qualified : j->rows;
        const uint32_t want = j->k < have ? j->k : have;
        for (uint32_t q = 0; q < j->nq; ++q)
            if (j->h_status[1 + q] != want) {
                ok = false;
                j->failed.push_back(q);
            }
    }
    if (j->observed || plan == kPredict) {
        pg_table* tm = const_cast<pg_table*>(j->t);
        std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
        if (j->observed && tm->pred_k == j->k) {
            double st[5];
            memcpy(st, j->h_status + kPredStatsAt, sizeof st);
            if (st[4] >= tm->pred_total) {              // (two streams report: keep the later snapshot)
                tm->pred_total = st[4];
                tm->pred_n = st[0];
                tm->pred_sum = st[1];
                tm->pred_sum2 = st[2];
                tm->pred_min = st[3];
            }
        }
        if (ctx->knobs.debug_scan && tm->pred_n > 0.0) {
            const double mean = tm->pred_sum / tm->pred_n, var = tm->pred_sum2 / tm->pred_n - mean * mean;
            fprintf(stderr, "[pg] plan %d %s: threshold model n = %.0f, z mean %.5f sd %.5f min %.5f (z_lo of this job %.5f)\n", plan,
                    ok ? "held" : "FAILED", tm->pred_n, mean, var > 0 ? sqrt(var) : 0.0, tm->pred_min, j->z_lo);
        }
        if (plan == kPredict && !ok) {
            // the model's threshold was too high for some query: this table stays on the pilot plan for a while (the
            // whole batch re-runs when more than a handful failed — that must stay rare)
            // and the model starts over: what it learnt no longer describes the queries
            tm->pred_failures++;
            tm->pred_backoff = 64u << (tm->pred_failures < 6 ? tm->pred_failures : 6);
            PG_HIP(hipMemsetAsync(tm->d_pred + 128 + 128 * 128, 0, 24, ctx->stream));
            tm->pred_n = tm->pred_sum = tm->pred_sum2 = 0.0;
        }
    }
    if (j->screen) {
        // A screened plan that OVERFLOWED (not: a threshold a few queries' lists fell short of) says the rows crowd within the
        // screen's error of the K-th scores — nearly collinear rows and queries along them: 1 % of 40 M rows within the int8
        // margin of every query's threshold — and the chunked fallback plans would screen the same crowd chunk by chunk (256
        // queries, 40 M such rows: 180 ms shuffled, 790 ms in ascending order, where the exact scan takes 30).  The job's
        // remaining plans run on the exact scan; two such batches in a row and the table's next 64 start there.
        pg_table* tm = const_cast<pg_table*>(j->t);
        std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
        if (j->h_status[0] != 0 && j->h_status[kI4mStatAt + 2] != 0 && (tm->rec_scale ? tm->rec_scale : 1) < ctx->knobs.max_rec_scale) {
            // the hit-record areas of the 256-query pass were too small for this table (clustered rows: a query's whole cluster sits
            // within the screen's error of its K-th score — tens of suspects per answer where uniform rows have two): they grow, and
            // the same plan runs again; the table keeps the larger areas
            tm->rec_scale = tm->rec_scale ? tm->rec_scale * 2 : 2;
            ctx->stats.recall_record_growths++;
            if (j->next_plan > 0) j->next_plan--;
            if (ctx->knobs.debug_scan) fprintf(stderr, "[pg] plan %d overflowed its hit-record areas: scale -> %u, again\n", plan, tm->rec_scale);
        } else if (j->h_status[0] != 0) {
            ctx->stats.recall_screen_overflows++;
            j->screen = j->screen4 = j->screen4m = j->l2_per_row = false;
            j->pred_observe = false;
            // (... starting over at the pilot plan: on the exact scan the sample's threshold is as good as anywhere, while the
            //  growing-chunk plans meet a table in ascending order with a flood of candidates per chunk)
            for (int p = 0; p < j->n_plans; ++p)
                if (j->plans[p] == kPilot && p < j->next_plan) j->next_plan = p;
            if (++tm->screen_overflow_streak >= 2) tm->screen_backoff = 64;
        } else if (ok) {
            tm->screen_overflow_streak = 0;
        }
    }
    if (j->susp_stat && ok && j->stat_wide) {
        // (the int8 screen's suspects per query: decides whether the table's passes get the refinement stage)
        pg_table* tm = const_cast<pg_table*>(j->t);
        std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
        const float per_q = (float)j->h_status[kI4mStatAt + 1] / (float)j->nq;
        tm->wide_susp = tm->wide_susp > 0.0f ? 0.75f * tm->wide_susp + 0.25f * per_q : per_q;
    }
    if (plan == kPredict && ok && j->susp_stat &&
        (double)j->h_status[kI4mStatAt + 4] > ctx->knobs.predict_max_factor * (double)j->k * j->nq) {
        // The predicted thresholds held but admitted far more than K rows per query as CANDIDATES (exact score >= threshold): on
        // clustered rows the model's margin — a few per cent of the distance between a query's mean score and its K-th best — is
        // many times the spread of the scores inside the query's cluster, every member of which then is a true candidate (20 per
        // answer; the refinement stage cannot reject what really reaches the threshold, while the sample's threshold leaves 7-15
        // suspects of which it keeps one).  Such a table goes back to the pilot plan; the model gets another try after an
        // exponentially growing number of batches.
        pg_table* tm = const_cast<pg_table*>(j->t);
        std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
        tm->pred_failures++;
        tm->pred_backoff = 64u << (tm->pred_failures < 6 ? tm->pred_failures : 6);
        if (ctx->knobs.debug_scan) fprintf(stderr, "[pg] plan %d held with %.1f candidates per answer: pilot plan for the next %u batches\n", plan,
                                           (double)j->h_status[kI4mStatAt + 4] / j->nq / j->k, tm->pred_backoff);
    }
    if (j->susp_stat && ok) {
        ctx->stats.recall_rescored += j->h_status[kI4mStatAt + 3];
        ctx->stats.recall_suspects += j->h_status[kI4mStatAt + 1];
        ctx->stats.recall_suspect_queries += j->nq;
        ctx->stats.recall_i4m_pairs += j->h_status[kI4mStatAt];
    }
    if (j->screen4m && j->susp_stat && ok) {
        // what the 4-bit stage lets through per query decides up to which batch size it beats the int8 shadow (recall_job_prepare)
        pg_table* tm = const_cast<pg_table*>(j->t);
        std::lock_guard<std::mutex> build_guard(g_stats_build_mu);
        const float per_q = (float)j->h_status[kI4mStatAt] / (float)j->nq;
        tm->i4m_pairs = tm->i4m_pairs > 0.0f ? 0.75f * tm->i4m_pairs + 0.25f * per_q : per_q;
        if (ctx->knobs.debug_scan) fprintf(stderr, "[pg] plan %d 4-bit stage: %.0f pairs per query (running %.0f)\n", plan, per_q, tm->i4m_pairs);
    }
    if (!ok) ctx->stats.recall_rescans++;
    if (ok && plan == kPredict) ctx->stats.recall_predicted++;
    // a refined threshold that fails verification on most of a batch says the head of the table is not representative
    // (ordered rows): after two such batches the table's recalls stop refining (a hint, not state anyone relies on)
    if (!ok && (plan == kPilot || plan == kPredict) && j->refined && j->failed.size() > j->nq / 2) const_cast<pg_table*>(j->t)->prefix_failures++;
    *ok_out = ok;
    return PG_OK;
}

void recall_job_finish(RecallJob* j) {
    pg_ctx* ctx = j->ctx;
    ctx->stats.recall_calls++;
    ctx->stats.recall_rows_scanned += j->scanned_rows;
    if (j->timers) {                                   // (a batch without stage timers leaves the last direct call's figures)
        ctx->stats.last_recall_ms = j->total_ms;
        ctx->last_scan_ms = j->scan_ms;
    }
    ctx->last_scan_launches = j->scan_launches;
    ctx->last_scan_bytes = j->scan_bytes;              // bytes the scan launches streamed (fp32 rows, int8 / bf16 / 4-bit shadow)
}

int recall_patch_failed_locked(RecallJob* j, uint32_t* counts) {
    pg_ctx* ctx = j->ctx;
    const std::vector<uint32_t> failed = j->failed;
    for (uint32_t q : failed) {
        uint32_t cnt = 0;
        int rc;
        if ((rc = recall_dev_locked(ctx, j->t, j->d_queries + (size_t)q * j->t->dim, 1, j->k, j->d_out_rows + (size_t)q * j->k,
                                    j->d_out_scores + (size_t)q * j->k, &cnt, j->d_out_count ? j->d_out_count + q : nullptr, true, j->l2,
                                    j->filter.col ? &j->filter : nullptr, j->exact_only)))
            return rc;
        counts[q] = cnt;
    }
    j->failed.clear();
    return PG_OK;
}

// the whole recall for one batch of queries, verified before it returns; all pointers are device pointers
int recall_dev_locked(pg_ctx* ctx, const pg_table* t, const float* d_queries, uint32_t nq,
                      uint32_t k, uint64_t* d_out_rows, float* d_out_scores,
                      uint32_t* out_count, uint32_t* d_out_count, bool skip_pilot, bool l2, const RowFilter* filter, bool exact_only) {
    RecallJob j;
    j.exact_only = exact_only;
    j.skip_pilot = skip_pilot;
    j.l2 = l2;
    if (filter) j.filter = *filter;
    j.ctx = ctx;
    j.t = t;
    j.d_queries = d_queries;
    j.nq = nq;
    j.k = k;
    j.d_out_rows = d_out_rows;
    j.d_out_scores = d_out_scores;
    j.d_out_count = d_out_count;
    j.h_status = ctx->h_status;
    j.events = &ctx->ev_pool;
    int rc;
    if ((rc = recall_job_prepare(&j))) return rc;
    uint32_t counts[kMaxQueries];
    for (;;) {
        if ((rc = recall_job_enqueue(&j))) return rc;
        PG_HIP(hipStreamSynchronize(ctx->stream));
        bool ok = false;
        if ((rc = recall_job_check(&j, &ok))) return rc;
        for (uint32_t q = 0; q < nq; ++q) counts[q] = ctx->h_status[1 + q];
        if (ok) break;
        if (!j.failed.empty() && j.failed.size() <= kMaxPatchQueries && nq > 1) {
            // the pilot's threshold was too high for a few queries only: re-run those, not the batch
            // (the nested calls reuse ctx->h_status: the counts are kept on the stack)
            if ((rc = recall_patch_failed_locked(&j, counts))) return rc;
            break;
        }
    }
    if (out_count)
        for (uint32_t q = 0; q < nq; ++q) out_count[q] = counts[q];
    recall_job_finish(&j);
    return PG_OK;
}

int topk_merge_strided_locked(pg_ctx* ctx, const uint64_t* d_rows, const float* d_scores, uint32_t nq, uint32_t nlists,
                              uint32_t per_list, size_t row_ls, size_t row_qs, size_t sc_ls, size_t sc_qs, uint32_t k,
                              uint64_t* d_out_rows, float* d_out_scores, uint32_t* d_out_count) {
    if (nq < 1 || nq > (uint32_t)kMaxQueries) {
        set_error("topk merge: nq=%u out of range", nq);
        return PG_ERR_INVALID;
    }
    const uint64_t per_q = (uint64_t)nlists * per_list;
    if (k < 1 || k > 16384 || per_q == 0 || per_q > k + kCandSlack) {
        set_error("topk merge: unsupported sizes k=%u lists=%u x %u", k, nlists, per_list);
        return PG_ERR_UNSUPPORTED;
    }
    RecallScratch rs;
    int rc;
    if ((rc = recall_scratch(ctx, 64, k, &rs))) return rc;
    PG_HIP(hipMemsetAsync(rs.cnt, 0, sizeof(uint32_t) * kMaxQueries, ctx->stream));
    dim3 grid((uint32_t)((per_q + 255) / 256), nq);
    merge_keys_kernel<<<grid, 256, 0, ctx->stream>>>(d_rows, d_scores, nq, nlists, per_list, row_ls, row_qs, sc_ls, sc_qs, rs.cap, rs.cand[0],
                                                    rs.cnt);
    PG_HIP(hipGetLastError());
    if ((rc = launch_select(ctx, nq, rs.cand[0], rs.cand[1], rs.cnt, rs.thr, rs.cap, k, 0))) return rc;
    return final_launch(ctx, rs.cand[1], rs.cnt, rs.cap, nq, k, 0, d_out_rows, d_out_scores, d_out_count);
}

// global top-k of nlists per-shard lists per query (identical and deterministic on every shard that runs it)
int topk_merge_locked(pg_ctx* ctx, const uint64_t* d_rows, const float* d_scores, uint32_t nq, uint32_t nlists,
                      uint32_t per_list, int list_major, uint32_t k, uint64_t* d_out_rows, float* d_out_scores,
                      uint32_t* d_out_count) {
    const size_t ls = list_major ? (size_t)nq * per_list : (size_t)per_list;
    const size_t qs = list_major ? (size_t)per_list : (size_t)nlists * per_list;
    return topk_merge_strided_locked(ctx, d_rows, d_scores, nq, nlists, per_list, ls, qs, ls, qs, k, d_out_rows, d_out_scores, d_out_count);
}

}  // namespace pg

extern "C" {

int pg_table_screen_info(pg_ctx* ctx, const pg_table* t, int* out_elem_bytes, float* out_scale, float* out_resid) {
    PG_REQUIRE(ctx && t, "pg_table_screen_info: NULL argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    int rc;
    if ((rc = pg::ensure_table_stats(ctx, t))) return rc;
    const bool screened = t->stats_valid && t->all_finite;
    if (out_elem_bytes) *out_elem_bytes = screened ? (t->shadow_is_i8 ? 1 : 2) : 0;
    if (out_scale) *out_scale = screened && t->shadow_is_i8 ? t->s8 : 0.0f;
    if (out_resid) *out_resid = screened && t->shadow_is_i8 ? t->resid8 : 0.0f;
    return PG_OK;
}

int pg_recall_topk_dev(pg_ctx* ctx, const pg_table* t, const float* d_queries, uint32_t nq,
                       uint32_t k, uint64_t* d_out_rows, float* d_out_scores, uint32_t* out_count) {
    PG_REQUIRE(ctx && t && d_queries && d_out_rows && d_out_scores, "pg_recall_topk_dev: NULL argument");
    PG_REQUIRE(nq >= 1 && nq <= (uint32_t)pg::kMaxQueries, "pg_recall_topk_dev: nq=%u must be in [1,%d]", nq, pg::kMaxQueries);
    PG_REQUIRE(t->dim <= 128 || nq <= 32, "pg_recall_topk_dev: dim %u supports at most 32 queries per call", t->dim);
    if (k < 1 || k > 16384) {
        pg::set_error("pg_recall_topk_dev: k=%u unsupported (1..16384)", k);
        return PG_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    return pg::recall_dev_locked(ctx, t, d_queries, nq, k, d_out_rows, d_out_scores, out_count, nullptr);
}

int pg_recall_topk(pg_ctx* ctx, const pg_table* t, const float* queries, uint32_t nq, uint32_t k,
                   uint64_t* out_rows, float* out_scores, uint32_t* out_count) {
    PG_REQUIRE(ctx && t && queries && out_rows && out_scores, "pg_recall_topk: NULL argument");
    PG_REQUIRE(nq >= 1 && nq <= (uint32_t)pg::kMaxQueries, "pg_recall_topk: nq=%u must be in [1,%d]", nq, pg::kMaxQueries);
    PG_REQUIRE(t->dim <= 128 || nq <= 32, "pg_recall_topk: dim %u supports at most 32 queries per call", t->dim);
    if (k < 1 || k > 16384) {
        pg::set_error("pg_recall_topk: k=%u unsupported (1..16384)", k);
        return PG_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    void* buf;
    int rc;
    const size_t qb = (size_t)nq * t->dim * 4, rb = (size_t)nq * k * 8, sb = (size_t)nq * k * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, qb + rb + sb + 64, &buf))) return rc;
    float* d_q = (float*)buf;
    uint64_t* d_rows = (uint64_t*)((char*)buf + ((qb + 15) & ~(size_t)15));
    float* d_sc = (float*)((char*)d_rows + rb);
    PG_HIP(hipMemcpyAsync(d_q, queries, qb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::recall_dev_locked(ctx, t, d_q, nq, k, d_rows, d_sc, out_count, nullptr))) return rc;
    PG_HIP(hipMemcpyAsync(out_rows, d_rows, rb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(out_scores, d_sc, sb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

// HologresVectorRecallV2 (service/recall/hologres_vector_recall_v2.go:23,96-206): "SELECT id, pm_approx_squared_euclidean_distance
// (emb, $1) as distance … ORDER BY distance LIMIT n" — the K rows of SMALLEST squared Euclidean distance, ascending, the distance
// as the item's score (:181-189).  Exact here (the reference's Proxima index is approximate): d = fmaf(-2, ip, |x|^2 + |q|^2),
// every sum a k-ascending fp32 fmaf chain; ties by row ascending; slots beyond the table's rows: row UINT64_MAX, distance +inf.
int pg_recall_topk_l2_dev(pg_ctx* ctx, const pg_table* t, const float* d_queries, uint32_t nq, uint32_t k, uint64_t* d_out_rows,
                          float* d_out_dist, uint32_t* out_count) {
    PG_REQUIRE(ctx && t && d_queries && d_out_rows && d_out_dist, "pg_recall_topk_l2_dev: NULL argument");
    PG_REQUIRE(nq >= 1 && nq <= (uint32_t)pg::kMaxQueries, "pg_recall_topk_l2_dev: nq=%u must be in [1,%d]", nq, pg::kMaxQueries);
    if (k < 1 || k > 16384) {
        pg::set_error("pg_recall_topk_l2_dev: k=%u unsupported (1..16384)", k);
        return PG_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    // (the screened pass serves up to 128 queries; the exact scan runs groups of 64 either way: at most 128 per job)
    for (uint32_t q0 = 0; q0 < nq; q0 += 128) {
        const uint32_t n = nq - q0 < 128 ? nq - q0 : 128;
        const int rc = pg::recall_dev_locked(ctx, t, d_queries + (size_t)q0 * t->dim, n, k, d_out_rows + (size_t)q0 * k,
                                             d_out_dist + (size_t)q0 * k, out_count ? out_count + q0 : nullptr, nullptr, false, true);
        if (rc) return rc;
    }
    return PG_OK;
}

int pg_recall_topk_l2(pg_ctx* ctx, const pg_table* t, const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows,
                      float* out_dist, uint32_t* out_count) {
    PG_REQUIRE(ctx && t && queries && out_rows && out_dist, "pg_recall_topk_l2: NULL argument");
    PG_REQUIRE(nq >= 1 && nq <= (uint32_t)pg::kMaxQueries, "pg_recall_topk_l2: nq=%u must be in [1,%d]", nq, pg::kMaxQueries);
    if (k < 1 || k > 16384) {
        pg::set_error("pg_recall_topk_l2: k=%u unsupported (1..16384)", k);
        return PG_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    void* buf;
    int rc;
    const size_t qb = (size_t)nq * t->dim * 4, rb = (size_t)nq * k * 8, sb = (size_t)nq * k * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, qb + rb + sb + 64, &buf))) return rc;
    float* d_q = (float*)buf;
    uint64_t* d_rows = (uint64_t*)((char*)buf + ((qb + 15) & ~(size_t)15));
    float* d_sc = (float*)((char*)d_rows + rb);
    PG_HIP(hipMemcpyAsync(d_q, queries, qb, hipMemcpyHostToDevice, ctx->stream));
    for (uint32_t q0 = 0; q0 < nq; q0 += 128) {
        const uint32_t n = nq - q0 < 128 ? nq - q0 : 128;
        if ((rc = pg::recall_dev_locked(ctx, t, d_q + (size_t)q0 * t->dim, n, k, d_rows + (size_t)q0 * k, d_sc + (size_t)q0 * k,
                                        out_count ? out_count + q0 : nullptr, nullptr, false, true)))
            return rc;
    }
    PG_HIP(hipMemcpyAsync(out_rows, d_rows, rb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(out_dist, d_sc, sb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

// A Hologres vector recall WITH its WhereClause (HologresVectorConf.WhereClause, recconf.go:492-497; the SQL of
// hologres_vector_recall.go:23 / hologres_vector_recall_v2.go:23 has `FROM table WHERE … ORDER BY distance`), in the shape the
// device serves: `column OP constant` over an integer feature column keyed by item row (e.g. "create_time > ${time}" with the
// constant substituted by the caller, as :56-61 does).  metric 0: inner product, descending; 1: squared Euclidean, ascending.
// Only rows that pass are candidates; out_count[q] = min(k, rows that pass).  Exact, like the unfiltered recalls: the predicate
// is applied where candidates are made, so the plans' thresholds are thresholds of the filtered top-K.
int pg_recall_topk_where(pg_ctx* ctx, const pg_table* t, const pg_features* fs, int column, int op, long long value, int metric,
                         const float* queries, uint32_t nq, uint32_t k, uint64_t* out_rows, float* out_scores, uint32_t* out_count) {
    PG_REQUIRE(ctx && t && fs && queries && out_rows && out_scores, "pg_recall_topk_where: NULL argument");
    PG_REQUIRE(nq >= 1 && nq <= (uint32_t)pg::kMaxQueries, "pg_recall_topk_where: nq=%u must be in [1,%d]", nq, pg::kMaxQueries);
    PG_REQUIRE(t->dim <= 128 || nq <= 32, "pg_recall_topk_where: dim %u supports at most 32 queries per call", t->dim);
    PG_REQUIRE(column >= 0 && (size_t)column < fs->cols.size(), "pg_recall_topk_where: column %d out of range", column);
    PG_REQUIRE(op >= 0 && op <= 5, "pg_recall_topk_where: op %d unknown (0 >, 1 >=, 2 <, 3 <=, 4 ==, 5 !=)", op);
    PG_REQUIRE(metric == 0 || metric == 1, "pg_recall_topk_where: metric %d unknown (0 inner product, 1 squared Euclidean)", metric);
    PG_REQUIRE(fs->rows >= t->rows, "pg_recall_topk_where: the feature store holds %llu rows, the table %llu",
               (unsigned long long)fs->rows, (unsigned long long)t->rows);
    const pg_features::Column& c = fs->cols[(size_t)column];
    if ((c.dtype != PG_F_I32 && c.dtype != PG_F_I64) || !c.d) {
        pg::set_error("pg_recall_topk_where: column \"%s\" must be an int32 / int64 column with values", c.name.c_str());
        return PG_ERR_UNSUPPORTED;
    }
    if (k < 1 || k > 16384) {
        pg::set_error("pg_recall_topk_where: k=%u unsupported (1..16384)", k);
        return PG_ERR_UNSUPPORTED;
    }
    pg::RowFilter f;
    f.col = c.d;
    f.dtype = c.dtype;
    f.op = op;
    f.val = value;
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    void* buf;
    int rc;
    const size_t qb = (size_t)nq * t->dim * 4, rb = (size_t)nq * k * 8, sb = (size_t)nq * k * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, qb + rb + sb + 64, &buf))) return rc;
    float* d_q = (float*)buf;
    uint64_t* d_rows = (uint64_t*)((char*)buf + ((qb + 15) & ~(size_t)15));
    float* d_sc = (float*)((char*)d_rows + rb);
    PG_HIP(hipMemcpyAsync(d_q, queries, qb, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *d_blk, *d_grp, cblocks, admitted;
    if ((rc = pg::filter_count_locked(ctx, f, t->rows, &d_blk, &d_grp, &cblocks, &admitted))) return rc;
    f.admitted = admitted;
    const uint32_t step = metric == 1 ? 128u : (uint32_t)pg::kMaxQueries;
    if (admitted == 0) {
        // nothing passes: every slot is padding (row UINT64_MAX, score -inf / distance +inf), every count 0
        pg::recall_pad_kernel<<<(uint32_t)(((size_t)nq * k + 255) / 256), 256, 0, ctx->stream>>>(d_rows, d_sc, (size_t)nq * k, metric == 1);
        PG_HIP(hipGetLastError());
        if (out_count) for (uint32_t q = 0; q < nq; ++q) out_count[q] = 0;
    } else if (admitted <= (nq <= 4 ? ctx->knobs.where_compact_max_rows / 2 : ctx->knobs.where_compact_max_rows) &&
               (uint64_t)admitted * ctx->knobs.where_compact_min_ratio <= t->rows &&
               (metric == 0 || t->dim == 64 || t->dim == 128)) {
        // A selective filter: the thresholds the scan plans estimate from samples of the table say little about the few rows
        // that pass (1 % admitted of 100 M rows, 128 queries: 180 ms of re-planned passes).  Gather the admitted rows, in row
        // order, into a compact table and run the exact scan over that: its rows are the candidates, its row order the tie
        // order, and the answer's local rows map back through the id list.  (The gather is 1 KB of traffic per admitted row: at
        // 100 M x 128 the copy wins below ~4 M rows for a lone query — in place 2.2 ms at 4 % — and below ~8 M for 16+ queries.)
        const size_t idb = (((size_t)admitted + 64) * 4 + 255) & ~(size_t)255;
        const size_t tabb = (((size_t)admitted + 64) * t->dim * 4 + 255) & ~(size_t)255;
        const size_t nxb = metric == 1 ? (((size_t)admitted + 64) * 4 + 255) & ~(size_t)255 : 0;
        void* gbuf;
        if ((rc = pg::scratch_reserve(ctx, 13, idb + tabb + nxb, &gbuf))) return rc;
        uint32_t* d_ids = (uint32_t*)gbuf;
        float* d_tab = (float*)((char*)gbuf + idb);
        float* d_cnx = nxb ? (float*)((char*)gbuf + idb + tabb) : nullptr;
        pg::filter_scatter_kernel<<<cblocks, 256, 0, ctx->stream>>>(f, t->rows, d_blk, d_grp, d_ids);
        const uint64_t quads = (uint64_t)admitted * (t->dim / 4);
        pg::compact_gather_kernel<<<(uint32_t)((quads + 255) / 256), 256, 0, ctx->stream>>>(t->d, d_ids, admitted, t->dim, d_tab);
        PG_HIP(hipMemsetAsync(d_tab + (size_t)admitted * t->dim, 0, (size_t)64 * t->dim * 4, ctx->stream));
        pg_table ct;
        ct.d = d_tab;
        ct.rows = admitted;
        ct.dim = t->dim;
        if (metric == 1) {
            if ((rc = pg::ensure_table_nx(ctx, t))) return rc;
            pg::compact_gather_nx_kernel<<<(admitted + 64 + 255) / 256, 256, 0, ctx->stream>>>(t->d_nx, d_ids, admitted, d_cnx);
            ct.d_nx = d_cnx;
            ct.nx_valid = true;
        }
        PG_HIP(hipGetLastError());
        for (uint32_t q0 = 0; q0 < nq; q0 += step) {
            const uint32_t n = nq - q0 < step ? nq - q0 : step;
            if ((rc = pg::recall_dev_locked(ctx, &ct, d_q + (size_t)q0 * t->dim, n, k, d_rows + (size_t)q0 * k, d_sc + (size_t)q0 * k,
                                            out_count ? out_count + q0 : nullptr, nullptr, false, metric == 1, nullptr, true)))
                return rc;
        }
        pg::compact_map_rows_kernel<<<(uint32_t)(((size_t)nq * k + 255) / 256), 256, 0, ctx->stream>>>(d_rows, (uint64_t)nq * k, d_ids, t->row_offset, admitted);
        PG_HIP(hipGetLastError());
        ct.d = nullptr;
        ct.d_nx = nullptr;
    } else {
        for (uint32_t q0 = 0; q0 < nq; q0 += step) {
            const uint32_t n = nq - q0 < step ? nq - q0 : step;
            if ((rc = pg::recall_dev_locked(ctx, t, d_q + (size_t)q0 * t->dim, n, k, d_rows + (size_t)q0 * k, d_sc + (size_t)q0 * k,
                                            out_count ? out_count + q0 : nullptr, nullptr, false, metric == 1, &f)))
                return rc;
        }
    }
    PG_HIP(hipMemcpyAsync(out_rows, d_rows, rb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(out_scores, d_sc, sb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

// A filtered VIEW of a table: the rows `column OP value` admits, in row order, copied into a table of their own whose recalls
// answer with the source's row ids.  What a Hologres recall with a WhereClause whose constant is fixed when the recall is built
// (hologres_vector_recall.go:56-61: "${time}" is substituted in the constructor) searches: build the view once per table
// generation and every recall flavour — pg_recall_topk[_l2][_dev], a coalescer's pg_coalescer_recall[_l2] — serves it at the speed
// of an unfiltered table of that size (own shadows, statistics and threshold model).  A snapshot: later changes of the source or
// the column do not reach it.  Ties break by source row, as in pg_recall_topk_where.  Destroyed with pg_table_destroy.
int pg_table_view_create(pg_ctx* ctx, const pg_table* t, const pg_features* fs, int column, int op, long long value, pg_table** out_view) {
    PG_REQUIRE(ctx && t && fs && out_view, "pg_table_view_create: NULL argument");
    PG_REQUIRE(!t->d_row_map, "pg_table_view_create: the source is a view itself");
    PG_REQUIRE(column >= 0 && (size_t)column < fs->cols.size(), "pg_table_view_create: column %d out of range", column);
    PG_REQUIRE(op >= 0 && op <= 5, "pg_table_view_create: op %d unknown (0 >, 1 >=, 2 <, 3 <=, 4 ==, 5 !=)", op);
    PG_REQUIRE(fs->rows >= t->rows, "pg_table_view_create: the feature store holds %llu rows, the table %llu", (unsigned long long)fs->rows,
               (unsigned long long)t->rows);
    const pg_features::Column& c = fs->cols[(size_t)column];
    if ((c.dtype != PG_F_I32 && c.dtype != PG_F_I64) || !c.d) {
        pg::set_error("pg_table_view_create: column \"%s\" must be an int32 / int64 column with values", c.name.c_str());
        return PG_ERR_UNSUPPORTED;
    }
    pg::RowFilter f;
    f.col = c.d;
    f.dtype = c.dtype;
    f.op = op;
    f.val = value;
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    PG_HIP(hipSetDevice(ctx->device));
    int rc;
    uint32_t *d_blk, *d_grp, cblocks, admitted;
    if ((rc = pg::filter_count_locked(ctx, f, t->rows, &d_blk, &d_grp, &cblocks, &admitted))) return rc;
    if (admitted == 0) {
        pg::set_error("pg_table_view_create: no row passes the filter");
        return PG_ERR_EMPTY;
    }
    pg_table* v = new pg_table();
    v->rows = admitted;
    v->dim = t->dim;
    v->row_offset = 0;
    v->map_offset = t->row_offset;
    hipError_t e = hipMalloc((void**)&v->d, ((size_t)admitted + 64) * t->dim * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&v->d_row_map, ((size_t)admitted + 64) * sizeof(uint32_t));
    if (e != hipSuccess) {
        pg::set_error("pg_table_view_create: hipMalloc(%.1f GB) failed: %s", (double)admitted * t->dim * 4 / 1e9, hipGetErrorString(e));
        if (v->d) (void)hipFree(v->d);
        delete v;
        return PG_ERR_NOMEM;
    }
    pg::filter_scatter_kernel<<<cblocks, 256, 0, ctx->stream>>>(f, t->rows, d_blk, d_grp, v->d_row_map);
    const uint64_t quads = (uint64_t)admitted * (t->dim / 4);
    pg::compact_gather_kernel<<<(uint32_t)((quads + 255) / 256), 256, 0, ctx->stream>>>(t->d, v->d_row_map, admitted, t->dim, v->d);
    PG_HIP(hipGetLastError());
    PG_HIP(hipMemsetAsync(v->d + (size_t)admitted * t->dim, 0, (size_t)64 * t->dim * sizeof(float), ctx->stream));
    PG_HIP(hipMemsetAsync(v->d_row_map + admitted, 0, 64 * sizeof(uint32_t), ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    *out_view = v;
    return PG_OK;
}

// I2IVectorRecall (service/recall/item_2_item_vector_racall.go:51-152): the trigger item's own embedding
// (dao.VectorString(item_id)) is the query of the same inner-product top-K; the trigger is not excluded (the
// reference's SQL does not exclude it either).
int pg_i2i_recall(pg_ctx* ctx, const pg_table* trigger_table, const uint32_t* trigger_rows, uint32_t n,
                  const pg_table* t, uint32_t k, uint64_t* out_rows, float* out_scores, uint32_t* out_count) {
    PG_REQUIRE(ctx && trigger_table && t && trigger_rows && out_rows && out_scores, "pg_i2i_recall: NULL argument");
    PG_REQUIRE(trigger_table->dim == t->dim, "pg_i2i_recall: trigger table dim %u != searched table dim %u", trigger_table->dim, t->dim);
    PG_REQUIRE(!trigger_table->d_row_map, "pg_i2i_recall: the trigger table is a filtered view (trigger rows are rows of the source)");
    PG_REQUIRE(n >= 1 && n <= (uint32_t)pg::kMaxQueries && (t->dim <= 128 || n <= 32), "pg_i2i_recall: %u trigger items per call unsupported", n);
    if (k < 1 || k > 16384) {
        pg::set_error("pg_i2i_recall: k=%u unsupported (1..16384)", k);
        return PG_ERR_UNSUPPORTED;
    }
    for (uint32_t i = 0; i < n; ++i)
        PG_REQUIRE(trigger_rows[i] < trigger_table->rows, "pg_i2i_recall: trigger row %u outside table of %llu rows", trigger_rows[i],
                   (unsigned long long)trigger_table->rows);
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead2 tr(t, trigger_table);
    void* buf;
    int rc;
    const size_t qb = (size_t)n * t->dim * 4, rb = (size_t)n * k * 8, sb = (size_t)n * k * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, qb + rb + sb + 64, &buf))) return rc;
    float* d_q = (float*)buf;
    uint64_t* d_rows = (uint64_t*)((char*)buf + ((qb + 15) & ~(size_t)15));
    float* d_sc = (float*)((char*)d_rows + rb);
    // the trigger rows are few: one small device-to-device copy each (coalesced 512-B rows)
    for (uint32_t i = 0; i < n; ++i)
        PG_HIP(hipMemcpyAsync(d_q + (size_t)i * t->dim, trigger_table->d + (size_t)trigger_rows[i] * t->dim, (size_t)t->dim * 4,
                              hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = pg::recall_dev_locked(ctx, t, d_q, n, k, d_rows, d_sc, out_count, nullptr))) return rc;
    PG_HIP(hipMemcpyAsync(out_rows, d_rows, rb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(out_scores, d_sc, sb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

// OnlineVectorRecall (service/recall/online_vector_recall.go:73-155): the model server turns the user's features
// into the user-tower embedding and answers with its FaissNeighNum nearest items (match_item_scores + item_ids,
// algorithm/eas/easyrec_response.go:700-734).  Here: user tower of the two-tower model on the device, then the
// same exact inner-product top-K over the item-embedding table (dim = the towers' output width).
int pg_online_vector_recall(pg_ctx* ctx, const pg_model* m, const pg_table* item_emb, const float* user_vecs,
                            uint32_t n_req, uint32_t k, uint64_t* out_rows, float* out_scores, uint32_t* out_count) {
    PG_REQUIRE(ctx && m && item_emb && user_vecs && out_rows && out_scores, "pg_online_vector_recall: NULL argument");
    PG_REQUIRE(m->kind == PG_MODEL_FM_TWOTOWER, "pg_online_vector_recall: model is not FM_TWOTOWER");
    PG_REQUIRE(item_emb->dim == m->to, "pg_online_vector_recall: item-embedding table dim %u != tower output %u", item_emb->dim, m->to);
    PG_REQUIRE(n_req >= 1 && n_req <= (uint32_t)pg::kMaxQueries, "pg_online_vector_recall: n_req=%u must be in [1,%d]", n_req, pg::kMaxQueries);
    if (k < 1 || k > 16384) {
        pg::set_error("pg_online_vector_recall: k=%u unsupported (1..16384)", k);
        return PG_ERR_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(item_emb->rw);
    void* buf;
    int rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t ub = al((size_t)n_req * m->d_user * 4), qb = al((size_t)n_req * m->to * 4), rb = (size_t)n_req * k * 8, sb = (size_t)n_req * k * 4;
    if ((rc = pg::scratch_reserve(ctx, 5, ub + qb + rb + sb + 64, &buf))) return rc;
    float* d_u = (float*)buf;
    float* d_q = (float*)((char*)buf + ub);
    uint64_t* d_rows = (uint64_t*)((char*)buf + ub + qb);
    float* d_sc = (float*)((char*)d_rows + rb);
    PG_HIP(hipMemcpyAsync(d_u, user_vecs, (size_t)n_req * m->d_user * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::fm2t_user_embedding_locked(ctx, m, d_u, n_req, d_q))) return rc;
    if ((rc = pg::recall_dev_locked(ctx, item_emb, d_q, n_req, k, d_rows, d_sc, out_count, nullptr))) return rc;
    PG_HIP(hipMemcpyAsync(out_rows, d_rows, rb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipMemcpyAsync(out_scores, d_sc, sb, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_topk_merge_dev(pg_ctx* ctx, const uint64_t* d_rows, const float* d_scores, uint32_t nq,
                      uint32_t nlists, uint32_t per_list, uint32_t k, uint64_t* d_out_rows,
                      float* d_out_scores) {
    PG_REQUIRE(ctx && d_rows && d_scores && d_out_rows && d_out_scores, "pg_topk_merge_dev: NULL argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::topk_merge_locked(ctx, d_rows, d_scores, nq, nlists, per_list, 0, k, d_out_rows, d_out_scores, nullptr);
}

}  // extern "C"
