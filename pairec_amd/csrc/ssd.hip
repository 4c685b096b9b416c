// ssd.hip — SSD (sliding spectrum decomposition) diversity re-rank in fp64 on the device.
//
// Replaces SSDSort.SSDWithSlidingWindow (sort/ssd_sort.go:346-486, arXiv 2107.05204; built on gonum
// v0.12.0 floats.Dot / floats.Norm / mat.ScaleVec / stat.PopMeanVariance) and the per-item embedding
// treatment of loadEmbeddingCache (:246-252: parse → L2-normalise → append 1).
//
// Per pick t the reference (a) adds back the projection of the item that leaves the window to every
// unselected embedding, (b) projects the last pick out of every unselected embedding, (c) scores
// quality_j = r_j + volume·‖e_j‖ and takes the first maximum, (d) volume *= ‖e_pick‖.  Each of (a)–(c)
// is independent per candidate, so one pass over the embeddings per pick does all three: thread j walks
// candidate j's residual embedding twice (dot, then update + norm).  The embeddings live transposed in
// global memory ([k][candidate]: coalesced across threads, L2-resident), the pick's and the leaving
// item's embeddings are broadcast from LDS.
//
// Summation orders (gonum is not vendored in the reference tree, parity is unpinned at that boundary —
// DESIGN.md §5.7): dot = chain_{k asc} fma; norm = sqrt(chain fma); e ∓= p·f as separate multiply and
// add/subtract (ScaleVec, then floats.Sub/Add); quality = r + (volume·l2).  Bit-identical to
// oracle/oracle.c:orc_ssd_window.
#include "pipeline.hpp"
#include "bitonic_reg.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>

namespace pg {

// one thread per candidate: gather the fp32 row, widen, optionally L2-normalise (floats.Norm /
// floats.Scale(1/norm), ssd_sort.go:246-249), optionally append 1 (ensurePosSimilarity, :250-252);
// stored transposed Et[k][n]
__global__ void ssd_prepare_kernel(const float* __restrict__ tab, uint32_t tab_rows, uint32_t d,
                                   const uint32_t* __restrict__ cand, uint32_t n, int normalize,
                                   int append_one, double* __restrict__ Et) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cand += (size_t)blockIdx.y * n;                               // request blockIdx.y of a batch
    Et += (size_t)blockIdx.y * n * (d + (append_one ? 1u : 0u));
    uint32_t row = cand[i];
    row = row < tab_rows ? row : tab_rows - 1;
    const float* x = tab + (size_t)row * d;
    double inv = 1.0;
    if (normalize) {
        double ss = 0.0;
        for (uint32_t k = 0; k < d; ++k) {
            const double v = (double)x[k];
            ss = fma(v, v, ss);
        }
        inv = 1.0 / sqrt(ss);
    }
    for (uint32_t k = 0; k < d; ++k) {
        double v = (double)x[k];
        if (normalize) v = inv * v;
        Et[(size_t)k * n + i] = v;
    }
    if (append_one) Et[(size_t)d * n + i] = 1.0;
}

// floats.MaxIdx over v[0..n): first maximum, NaN skipped, all-NaN → 0.  Block-wide (wave shuffles, then one
// LDS round over the <= 16 wave winners); result in *out_idx.
__device__ __forceinline__ void ssd_block_argmax(const double* __restrict__ v, uint32_t n, double* s_val,
                                                 uint32_t* s_idx, uint32_t* out_idx) {
    const uint32_t tid = threadIdx.x;
    double best = 0.0;
    uint32_t bi = 0xFFFFFFFFu;                       // "none yet"
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const double x = v[i];
        if (x != x) continue;
        if (bi == 0xFFFFFFFFu || x > best) { best = x; bi = i; }
    }
    auto better = [](double ov, uint32_t oi, double mv, uint32_t mi) {
        return oi != 0xFFFFFFFFu && (mi == 0xFFFFFFFFu || ov > mv || (ov == mv && oi < mi));
    };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, 64);
        if (better(ov, oi, best, bi)) { best = ov; bi = oi; }
    }
    const uint32_t wave = tid >> 6, nw = blockDim.x >> 6;
    if ((tid & 63) == 0) { s_val[wave] = best; s_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        double mv = s_val[0];
        uint32_t mi = s_idx[0];
        for (uint32_t w = 1; w < nw; ++w)
            if (better(s_val[w], s_idx[w], mv, mi)) { mv = s_val[w]; mi = s_idx[w]; }
        *out_idx = (mi == 0xFFFFFFFFu) ? 0u : mi;
    }
    __syncthreads();
}

__device__ __forceinline__ bool ssd_bad(double x) { return x != x || fabs(x) == __builtin_inf(); }

constexpr uint32_t kSsdMaxDim = 320;
constexpr uint32_t kSsdBlk = 16;        // values fetched per batch (32 VGPRs)

// One workgroup.  Et: [d1][n] residual embeddings (modified), P: [window][n] ring of projection
// coefficients, nrm/q: [n], sel: [n] flags, out: [T] picks.
__global__ __launch_bounds__(1024) void ssd_kernel(double* __restrict__ Et, uint32_t n, uint32_t d1,
                                                   const double* __restrict__ rel, double gamma, uint32_t T,
                                                   uint32_t W, int star, double* __restrict__ P,
                                                   double* __restrict__ nrm, double* __restrict__ ssq,
                                                   double* __restrict__ q, uint32_t* __restrict__ sel, uint32_t* __restrict__ out) {
    __shared__ double s_val[16];
    __shared__ uint32_t s_idx[16];
    __shared__ uint32_t s_j;
    __shared__ double e_sel[kSsdMaxDim], e_old[kSsdMaxDim];
    __shared__ double s_den;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < n; i += blockDim.x) sel[i] = 0;
    __syncthreads();
    ssd_block_argmax(rel, n, s_val, s_idx, &s_j);
    uint32_t idx = s_j;
    if (tid == 0) { out[0] = idx; sel[idx] = 1; }
    // ‖e_first‖² (thread 0, once): volume = gamma · ‖e_first‖ (unless SSD*), and the first projection's
    // denominator.  Later denominators floats.Dot(e_pick, e_pick) are the pick's own sum of squares from the
    // pass that scored it — the same chain over the same values (a picked embedding is never modified).
    if (tid == 0) {
        double ss = 0.0;
        for (uint32_t k = 0; k < d1; ++k) {
            const double e = Et[(size_t)k * n + idx];
            ss = fma(e, e, ss);
        }
        s_den = ss;
    }
    __syncthreads();
    double den = s_den;
    double volume = gamma;
    if (!star && !ssd_bad(sqrt(den))) volume = __dmul_rn(volume, sqrt(den));

    for (uint32_t t = 1; t < T;) {
        const bool pop = t > W;
        const uint32_t i_old = pop ? out[t - 1 - W] : 0u;
        const uint32_t slot = t % W;
        for (uint32_t k = tid; k < d1; k += blockDim.x) {
            e_sel[k] = Et[(size_t)k * n + idx];
            if (pop) e_old[k] = Et[(size_t)k * n + i_old];
        }
        __syncthreads();
        for (uint32_t j = tid; j < n; j += blockDim.x) {
            if (sel[j]) { q[j] = -DBL_MAX; continue; }
            // The two k-chains are serial, the loads are not: full batches of kSsdBlk values are fetched
            // together (independent loads, all in flight — the residuals live in L2, not LDS) before the
            // chain runs over them; the tail (d1 mod kSsdBlk) goes element by element.
            double* col = Et + j;
            double acc = 0.0;
            const double pold = pop ? P[(size_t)slot * n + j] : 0.0;
            uint32_t k0 = 0;
            for (; k0 + kSsdBlk <= d1; k0 += kSsdBlk) {
                double v[kSsdBlk];
#pragma unroll
                for (uint32_t u = 0; u < kSsdBlk; ++u) v[u] = col[(size_t)(k0 + u) * n];
                if (pop) {
#pragma unroll
                    for (uint32_t u = 0; u < kSsdBlk; ++u) {
                        v[u] = __dadd_rn(v[u], __dmul_rn(pold, e_old[k0 + u]));
                        col[(size_t)(k0 + u) * n] = v[u];
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kSsdBlk; ++u) acc = fma(v[u], e_sel[k0 + u], acc);
            }
            for (; k0 < d1; ++k0) {
                double e = col[(size_t)k0 * n];
                if (pop) {
                    e = __dadd_rn(e, __dmul_rn(pold, e_old[k0]));
                    col[(size_t)k0 * n] = e;
                }
                acc = fma(e, e_sel[k0], acc);
            }
            double p = acc / den;
            if (ssd_bad(p)) p = 1.0;
            double ss = 0.0;
            for (k0 = 0; k0 + kSsdBlk <= d1; k0 += kSsdBlk) {
                double v[kSsdBlk];
#pragma unroll
                for (uint32_t u = 0; u < kSsdBlk; ++u) v[u] = col[(size_t)(k0 + u) * n];
#pragma unroll
                for (uint32_t u = 0; u < kSsdBlk; ++u) {
                    const double e = __dsub_rn(v[u], __dmul_rn(p, e_sel[k0 + u]));
                    col[(size_t)(k0 + u) * n] = e;
                    ss = fma(e, e, ss);
                }
            }
            for (; k0 < d1; ++k0) {
                const double e = __dsub_rn(col[(size_t)k0 * n], __dmul_rn(p, e_sel[k0]));
                col[(size_t)k0 * n] = e;
                ss = fma(e, e, ss);
            }
            const double l2 = sqrt(ss);
            P[(size_t)slot * n + j] = p;
            nrm[j] = l2;
            ssq[j] = ss;
            q[j] = __dadd_rn(rel[j], __dmul_rn(volume, ssd_bad(l2) ? 0.5 : l2));
        }
        __syncthreads();
        ssd_block_argmax(q, n, s_val, s_idx, &s_j);
        idx = s_j;
        ++t;
        if (tid == 0) { out[t - 1] = idx; sel[idx] = 1; }
        den = ssq[idx];
        if (!star) {
            const double l2 = nrm[idx];
            if (!ssd_bad(l2)) volume = __dmul_rn(volume, l2);
        }
        __syncthreads();
    }
}

// Register-resident variant for the common embedding widths: 256 threads (one wave per SIMD, 512 VGPRs
// each), a thread pulls a candidate's whole residual embedding into registers with D1 independent loads —
// one L2 round trip — runs restore, dot, update and norm on it and writes it back once; quality / norm /
// sum of squares / selection flags live in LDS.  Same arithmetic, same order as ssd_kernel.
constexpr uint32_t kSsdRegMaxN = 2048;
template <int D1>
__global__ __launch_bounds__(256) void ssd_kernel_reg(double* __restrict__ Et, uint32_t n,
                                                      const double* __restrict__ rel, double gamma, uint32_t T,
                                                      uint32_t W, int star, double* __restrict__ P,
                                                      uint32_t* __restrict__ out) {
    __shared__ double s_val[16];
    __shared__ uint32_t s_idx[16];
    __shared__ uint32_t s_j;
    __shared__ double e_sel[D1], e_old[D1];
    __shared__ double q_s[kSsdRegMaxN], nrm_s[kSsdRegMaxN], ssq_s[kSsdRegMaxN];
    __shared__ uint8_t sel_s[kSsdRegMaxN];
    __shared__ uint32_t out_s[kSsdRegMaxN];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < n; i += blockDim.x) { sel_s[i] = 0; q_s[i] = rel[i]; }
    __syncthreads();
    ssd_block_argmax(q_s, n, s_val, s_idx, &s_j);
    uint32_t idx = s_j;
    if (tid == 0) {
        out_s[0] = idx;
        sel_s[idx] = 1;
        double ss = 0.0;
        for (int k = 0; k < D1; ++k) {
            const double e = Et[(size_t)k * n + idx];
            ss = fma(e, e, ss);
        }
        ssq_s[idx] = ss;
    }
    __syncthreads();
    double den = ssq_s[idx];
    double volume = gamma;
    if (!star && !ssd_bad(sqrt(den))) volume = __dmul_rn(volume, sqrt(den));

    for (uint32_t t = 1; t < T;) {
        const bool pop = t > W;
        const uint32_t i_old = pop ? out_s[t - 1 - W] : 0u;
        const uint32_t slot = t % W;
        for (uint32_t k = tid; k < (uint32_t)D1; k += blockDim.x) {
            e_sel[k] = Et[(size_t)k * n + idx];
            if (pop) e_old[k] = Et[(size_t)k * n + i_old];
        }
        __syncthreads();
        for (uint32_t j = tid; j < n; j += blockDim.x) {
            if (sel_s[j]) { q_s[j] = -DBL_MAX; continue; }
            // (row base uniform, lane offset j: scalar-base addressing, no per-element address registers)
            const double pold = pop ? P[(size_t)slot * n + j] : 0.0;
            double v[D1];
#pragma unroll
            for (int k = 0; k < D1; ++k) v[k] = (Et + (size_t)k * n)[j];
            // (scheduling fences every 8 elements: without them hipcc hoists all 2·D1 LDS operand reads
            //  to the top and spills the embedding it was supposed to keep in registers)
            if (pop) {
#pragma unroll
                for (int k = 0; k < D1; ++k) {
                    v[k] = __dadd_rn(v[k], __dmul_rn(pold, e_old[k]));
                    if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D1; ++k) {
                acc = fma(v[k], e_sel[k], acc);
                if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            double p = acc / den;
            if (ssd_bad(p)) p = 1.0;
            double ss = 0.0;
#pragma unroll
            for (int k = 0; k < D1; ++k) {
                const double e = __dsub_rn(v[k], __dmul_rn(p, e_sel[k]));
                (Et + (size_t)k * n)[j] = e;
                ss = fma(e, e, ss);
                if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            const double l2 = sqrt(ss);
            P[(size_t)slot * n + j] = p;
            nrm_s[j] = l2;
            ssq_s[j] = ss;
            q_s[j] = __dadd_rn(rel[j], __dmul_rn(volume, ssd_bad(l2) ? 0.5 : l2));
        }
        __syncthreads();
        ssd_block_argmax(q_s, n, s_val, s_idx, &s_j);
        idx = s_j;
        ++t;
        if (tid == 0) { out_s[t - 1] = idx; sel_s[idx] = 1; }
        den = ssq_s[idx];
        if (!star) {
            const double l2 = nrm_s[idx];
            if (!ssd_bad(l2)) volume = __dmul_rn(volume, l2);
        }
        __syncthreads();
    }
    for (uint32_t i = tid; i < T; i += blockDim.x) out[i] = out_s[i];
}

// ---------------------------------------------------------------------------------------------
// Multi-workgroup variant: one wave per workgroup, one candidate per lane, the candidate's whole residual
// embedding lives in the lane's registers for the entire run (D1 doubles ≤ 258 VGPRs of the 512 a lone wave
// may use), so a pick moves no embedding data except through a global mailbox.  Workgroups meet at a device-wide
// barrier ONCE per pick (all of them are co-resident: at most 128 single-wave workgroups on 256 CUs): before it every
// workgroup publishes its own best candidate — quality, index, sum of squares, norm AND residual vector — and the
// owner of the item leaving the window publishes that item's frozen vector; after it every workgroup reduces the G
// partial results identically and reads the winner's vector from the winner's slot.  (Publishing only the global
// winner's vector needs a second barrier per pick — the first version: 10 us per pick, of which the two barriers
// were most.)  Same arithmetic, same order as ssd_kernel.
// ---------------------------------------------------------------------------------------------
struct SsdMail {
    uint32_t counter;          // monotonically increasing arrival count of the device-wide barrier
    uint32_t pad[15];
    double part_q[2][128];     // per-workgroup best quality, by pick parity
    uint32_t part_i[2][128];
    double part_ss[2][128];    // that candidate's sum of squares / norm
    double part_l2[2][128];
};

__device__ __forceinline__ double ld_dev(const double* p) {       // device-scope load (bypasses the CU's L1)
    return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_dev(const uint32_t* p) {
    return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void ssd_grid_barrier(uint32_t* counter, uint32_t G, uint32_t& phase) {
    ++phase;
    // (the other lanes' published stores are ordered before lane 0's release by the wave-scope fence: one wave per workgroup)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase * G)
            __builtin_amdgcn_s_sleep(1);
        // (no fence on the way out: everything read after the barrier is read with device-scope loads, ld_dev)
    }
    __builtin_amdgcn_wave_barrier();
}

// ebuf: [2 parities][G + 1 slots][D1]: slot w = workgroup w's best candidate, slot G = the item leaving the window
template <int D1>
__global__ __launch_bounds__(64) void ssd_kernel_grid(const double* __restrict__ Et, uint32_t n,
                                                      const double* __restrict__ rel_g, double gamma, uint32_t T,
                                                      uint32_t W, int star, SsdMail* __restrict__ mail,
                                                      double* __restrict__ ebuf, uint32_t* __restrict__ out) {
    __shared__ double e_sel[D1], e_old[D1];
    __shared__ double p_ring[16][64];
    __shared__ uint32_t pick_ring[32];                   // the last picks (W <= 16 back is all that is needed)
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x, G = gridDim.x, wg = blockIdx.x;
    {   // request blockIdx.y of a batch: its own embeddings, mailbox (and barrier counter), vector slots, output
        const uint32_t req = blockIdx.y;
        Et += (size_t)req * n * D1;
        rel_g += (size_t)req * n;
        mail += req;
        ebuf += (size_t)req * 2 * (G + 1) * D1;
        out += (size_t)req * T;
    }
    const uint32_t j = wg * 64 + lane;
    const bool valid = j < n;
    const uint32_t jc = valid ? j : n - 1;
    double v[D1];
#pragma unroll
    for (int k = 0; k < D1; ++k) v[k] = (Et + (size_t)k * n)[jc];
    const double rel = rel_g[jc];
    bool selected = false;
    uint32_t phase = 0;
    const double nan = __longlong_as_double(0x7FF8000000000000ll);

    double ss_own = 0.0, l2_own = 0.0;                  // this lane's last sum of squares / norm
    {   // ‖e‖² of every candidate once: the first pick's volume factor and first denominator
#pragma unroll
        for (int k = 0; k < D1; ++k) {
            ss_own = fma(v[k], v[k], ss_own);
            if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        l2_own = sqrt(ss_own);
    }
    double q = valid ? rel : nan;                        // quality of the coming pick ("first maximum, NaN skipped")
    uint32_t t = 1;
    double volume = gamma;
    for (;;) {
        const uint32_t par = t & 1;
        const bool pop = t > W && t < T;
        // ---- publish: this workgroup's best (with its vector), the item leaving the window
        // (argmax as in dpp_greedy_wave_kernel: the wave's maximum value through six lane exchanges, the first lane that holds it
        // from a ballot — instead of a butterfly over (value, index) pairs; the results are wave-uniform)
        double bq = wave_max_f64(valid ? q : nan);
        uint32_t bi = kNone;
        if (bq == bq) {
            const uint64_t bal = __builtin_amdgcn_ballot_w64(valid && q == bq);
            bi = wg * 64u + (uint32_t)__builtin_ctzll(bal);
        }
        if (lane == 0) { mail->part_q[par][wg] = bq; mail->part_i[par][wg] = bi; }
        // (nothing pickable anywhere → pick 0, as floats.MaxIdx does: workgroup 0 then publishes candidate 0)
        const uint32_t pub = bi != kNone ? bi : (wg == 0 ? 0u : kNone);
        if (valid && j == pub) {
            mail->part_ss[par][wg] = ss_own;
            mail->part_l2[par][wg] = l2_own;
            if (t < T) {
                double* dst = ebuf + ((size_t)par * (G + 1) + wg) * D1;
#pragma unroll
                for (int k = 0; k < D1; ++k) dst[k] = v[k];
            }
        }
        uint32_t i_old = kNone;
        if (pop) i_old = pick_ring[(t - 1 - W) & 31];
        if (pop && valid && j == i_old) {
            double* dst = ebuf + ((size_t)par * (G + 1) + G) * D1;
#pragma unroll
            for (int k = 0; k < D1; ++k) dst[k] = v[k];
        }
        ssd_grid_barrier(&mail->counter, G, phase);
        // ---- every workgroup reduces the G partial results identically
        // G <= 128 partial results, two slots per lane (workgroups `lane` and `lane + 64`; a workgroup's candidates precede the next
        // one's, so the lowest slot-then-lane that holds the maximum is the first maximum)
        double pq[2];
        uint32_t pi[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const uint32_t w = lane + 64u * (uint32_t)sl;
            pq[sl] = nan;
            pi[sl] = kNone;
            if (w < G) {
                pq[sl] = ld_dev(&mail->part_q[par][w]);
                pi[sl] = ld_dev(&mail->part_i[par][w]);
            }
            if (pi[sl] == kNone) pq[sl] = nan;
        }
        const double gq = wave_max_f64(fmax(pq[0], pq[1]));
        uint32_t gi = kNone;
        if (gq == gq) {
#pragma unroll
            for (int sl = 1; sl >= 0; --sl) {
                const uint64_t bal = __builtin_amdgcn_ballot_w64(pq[sl] == gq);
                if (bal) gi = (uint32_t)__builtin_amdgcn_readlane((int)pi[sl], (int)__builtin_ctzll(bal));
            }
        }
        const uint32_t idx = gi == kNone ? 0u : gi;      // pick number t
        const uint32_t wwg = idx >> 6;
        if (valid && j == idx) {
            selected = true;
            out[t - 1] = idx;
        }
        if (lane == 0) pick_ring[(t - 1) & 31] = idx;
        if (t >= T) break;
        {   // all the loads of the two vectors first (independent: one L2 round trip, not three)
            constexpr int NR = (D1 + 63) / 64;
            const double* ps = ebuf + ((size_t)par * (G + 1) + wwg) * D1;
            const double* po = ebuf + ((size_t)par * (G + 1) + G) * D1;
            double ts[NR], to[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const uint32_t k = lane + 64u * (uint32_t)r;
                ts[r] = k < (uint32_t)D1 ? ld_dev(ps + k) : 0.0;
                to[r] = (pop && k < (uint32_t)D1) ? ld_dev(po + k) : 0.0;
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const uint32_t k = lane + 64u * (uint32_t)r;
                if (k < (uint32_t)D1) { e_sel[k] = ts[r]; e_old[k] = to[r]; }
            }
        }
        const double den = ld_dev(&mail->part_ss[par][wwg]);
        const double l2p = ld_dev(&mail->part_l2[par][wwg]);
        if (!star && !ssd_bad(l2p)) volume = __dmul_rn(volume, l2p);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t slot = t % W;
        q = -DBL_MAX;
        if (!selected) {
            if (pop) {
                const double pold = p_ring[slot][lane];
#pragma unroll
                for (int k = 0; k < D1; ++k) {
                    v[k] = __dadd_rn(v[k], __dmul_rn(pold, e_old[k]));
                    if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            }
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < D1; ++k) {
                acc = fma(v[k], e_sel[k], acc);
                if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            double p = acc / den;
            if (ssd_bad(p)) p = 1.0;
            double ss = 0.0;
#pragma unroll
            for (int k = 0; k < D1; ++k) {
                v[k] = __dsub_rn(v[k], __dmul_rn(p, e_sel[k]));
                ss = fma(v[k], v[k], ss);
                if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            const double l2 = sqrt(ss);
            p_ring[slot][lane] = p;
            ss_own = ss;
            l2_own = l2;
            q = __dadd_rn(rel, __dmul_rn(volume, ssd_bad(l2) ? 0.5 : l2));
        }
        if (!valid) q = nan;
        ++t;
    }
}

// ssd_norm_quality_score (ssd_sort.go:360-388) on the caller's thread, in the reference's operation order
// (stat.PopMeanVariance two-pass with compensation, stat.StdScore; min-max with eps = 1e-6).  false: "all item score are
// zeros" — the reference returns the items unchanged.
bool ssd_norm_quality_host(const double* rel, uint32_t n, int mode, double* out) {
    if (mode == 1) {
        double sum = 0.0;
        for (uint32_t i = 0; i < n; ++i) sum = sum + rel[i];
        const double mean = sum / (double)n;
        double ss = 0.0, comp = 0.0;
        for (uint32_t i = 0; i < n; ++i) {
            const double d = rel[i] - mean;
            volatile double dd = d * d;
            ss = ss + dd;
            comp = comp + d;
        }
        volatile double cc = comp * comp;
        const double variance = (ss - cc / (double)n) / (double)n;
        if (mean == 0.0 || variance == 0.0) return false;
        const double sd = sqrt(variance);
        for (uint32_t i = 0; i < n; ++i) out[i] = (rel[i] - mean) / sd;
        return true;
    }
    if (mode == 2) {
        const double mx = rel[0], mn = rel[n - 1], span = mx - mn;
        if (span == 0.0) return false;
        const double eps = 1e-6;
        for (uint32_t i = 0; i < n; ++i) {
            volatile double a = ((rel[i] - mn) / span) * (1 - eps);
            out[i] = a + eps;
        }
        return true;
    }
    for (uint32_t i = 0; i < n; ++i) out[i] = rel[i];
    return true;
}

bool ssd_batchable(uint32_t d1, uint32_t window) { return (d1 == 64 || d1 == 65 || d1 == 128 || d1 == 129) && window <= 16; }

// SSDWithSlidingWindow for R requests of n candidates each, device-resident: d_cand [R][n] rows of `t`, d_rel [R][n]
// quality scores (normalised already), d_out [R][T] picks.  R > 1 needs the multi-workgroup kernel's shapes
// (ssd_batchable): every request is its own set of G single-wave workgroups with its own mailbox and barrier; a launch
// carries as many requests as are co-resident (4 single-wave workgroups per CU).  Caller holds ctx->mu.
int ssd_run_locked(pg_ctx* ctx, const pg_table* t, const uint32_t* d_cand, const double* d_rel, uint32_t R, uint32_t n,
                   double gamma, uint32_t T, uint32_t window, int normalize_emb, int ensure_pos_similarity, int use_ssd_star,
                   uint32_t* d_out) {
    if (R == 0 || n == 0 || T == 0) return PG_OK;
    if (window <= 1) window = 5;                          // ssd_sort.go:357-360
    const uint32_t d1 = t->dim + (ensure_pos_similarity ? 1u : 0u);
    const uint32_t G = (n + 63) / 64;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t bE = al((size_t)n * d1 * 8), bP = al((size_t)window * n * 8), bN = al((size_t)n * 8), bSel = al((size_t)n * 4);
    const size_t bMail = al(sizeof(SsdMail)), bEbuf = al((size_t)2 * (G + 1) * d1 * 8);
    // Kernel choice: the multi-workgroup kernel (embeddings pinned in registers, device-wide barrier) for the usual
    // widths (dim 64 / 128, with or without the appended 1) and windows <= 16; the one-workgroup kernels otherwise
    // (PG_SSD_KERNEL=generic|reg forces them, for A/B runs; single requests only).
    const char* force = getenv("PG_SSD_KERNEL");
    const bool known_d1 = d1 == 64 || d1 == 65 || d1 == 128 || d1 == 129;
    int kind = ssd_batchable(d1, window) ? 2 : ((known_d1 && n <= kSsdRegMaxN) ? 1 : 0);
    if (R == 1 && force && !strcmp(force, "generic")) kind = 0;
    if (R == 1 && force && !strcmp(force, "reg") && known_d1 && n <= kSsdRegMaxN) kind = 1;
    if (R > 1 && kind != 2) {
        set_error("ssd: a batch of %u requests needs dim 64 / 128 and a window <= 16 (d1 = %u, window %u)", R, d1, window);
        return PG_ERR_UNSUPPORTED;
    }
    void* buf;
    int rc;
    if (kind == 2) {
        if ((rc = scratch_reserve(ctx, 7, (size_t)R * (bE + bMail + bEbuf), &buf))) return rc;
        char* p = (char*)buf;
        double* Et = (double*)p; p += (size_t)R * bE;
        // (requests are strided by their exact sizes inside the kernels: n * d1 doubles, one SsdMail, 2 (G + 1) d1 doubles)
        SsdMail* mail = (SsdMail*)p; p += (size_t)R * bMail;
        double* ebuf = (double*)p;
        ssd_prepare_kernel<<<dim3((n + 63) / 64, R), 64, 0, ctx->stream>>>(t->d, (uint32_t)t->rows, t->dim, d_cand, n, normalize_emb,
                                                                           ensure_pos_similarity, Et);
        PG_HIP(hipMemsetAsync(mail, 0, (size_t)R * sizeof(SsdMail), ctx->stream));
        // as many requests per launch as are certainly co-resident: 4 single-wave workgroups (one per SIMD) per CU
        const uint32_t per_launch = std::max(1u, (uint32_t)ctx->num_cus * 4u / G);
        for (uint32_t r0 = 0; r0 < R; r0 += per_launch) {
            const uint32_t rn = std::min(per_launch, R - r0);
            const double* Et_r = Et + (size_t)r0 * n * d1;
            const double* rel_r = d_rel + (size_t)r0 * n;
            SsdMail* mail_r = mail + r0;
            double* ebuf_r = ebuf + (size_t)r0 * 2 * (G + 1) * d1;
            uint32_t* out_r = d_out + (size_t)r0 * T;
            const dim3 grid(G, rn);
            switch (d1) {
                case 64: ssd_kernel_grid<64><<<grid, 64, 0, ctx->stream>>>(Et_r, n, rel_r, gamma, T, window, use_ssd_star, mail_r, ebuf_r, out_r); break;
                case 65: ssd_kernel_grid<65><<<grid, 64, 0, ctx->stream>>>(Et_r, n, rel_r, gamma, T, window, use_ssd_star, mail_r, ebuf_r, out_r); break;
                case 128: ssd_kernel_grid<128><<<grid, 64, 0, ctx->stream>>>(Et_r, n, rel_r, gamma, T, window, use_ssd_star, mail_r, ebuf_r, out_r); break;
                default: ssd_kernel_grid<129><<<grid, 64, 0, ctx->stream>>>(Et_r, n, rel_r, gamma, T, window, use_ssd_star, mail_r, ebuf_r, out_r); break;
            }
        }
        PG_HIP(hipGetLastError());
        return PG_OK;
    }
    if ((rc = scratch_reserve(ctx, 7, bE + bP + 3 * bN + bSel, &buf))) return rc;
    char* p = (char*)buf;
    double* Et = (double*)p; p += bE;
    double* P = (double*)p; p += bP;
    double* nrm = (double*)p; p += bN;
    double* ssq = (double*)p; p += bN;
    double* q = (double*)p; p += bN;
    uint32_t* d_sel = (uint32_t*)p;
    ssd_prepare_kernel<<<dim3((n + 63) / 64, 1), 64, 0, ctx->stream>>>(t->d, (uint32_t)t->rows, t->dim, d_cand, n, normalize_emb,
                                                                       ensure_pos_similarity, Et);
    if (kind == 1) {
        switch (d1) {
            case 64: ssd_kernel_reg<64><<<1, 256, 0, ctx->stream>>>(Et, n, d_rel, gamma, T, window, use_ssd_star, P, d_out); break;
            case 65: ssd_kernel_reg<65><<<1, 256, 0, ctx->stream>>>(Et, n, d_rel, gamma, T, window, use_ssd_star, P, d_out); break;
            case 128: ssd_kernel_reg<128><<<1, 256, 0, ctx->stream>>>(Et, n, d_rel, gamma, T, window, use_ssd_star, P, d_out); break;
            default: ssd_kernel_reg<129><<<1, 256, 0, ctx->stream>>>(Et, n, d_rel, gamma, T, window, use_ssd_star, P, d_out); break;
        }
    } else {
        ssd_kernel<<<1, 1024, 0, ctx->stream>>>(Et, n, d1, d_rel, gamma, T, window, use_ssd_star, P, nrm, ssq, q, d_sel, d_out);
    }
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // namespace pg

extern "C" {

int pg_ssd(pg_ctx* ctx, const pg_table* t, const uint32_t* cand_rows, const double* rel, uint32_t n,
           double gamma, uint32_t topn, uint32_t window, int normalize_emb, int ensure_pos_similarity,
           int norm_quality_score, int use_ssd_star, uint32_t* out_idx, uint32_t* out_count,
           double* out_quality) {
    PG_REQUIRE(ctx && t && out_count, "pg_ssd: NULL argument");
    *out_count = 0;
    if (n == 0 || topn == 0) return PG_OK;
    PG_REQUIRE(cand_rows && rel && out_idx, "pg_ssd: NULL argument");
    PG_REQUIRE(norm_quality_score >= 0 && norm_quality_score <= 2, "pg_ssd: norm_quality_score must be 0, 1 or 2");
    if (window <= 1) window = 5;                          // ssd_sort.go:357-360
    const uint32_t d1 = t->dim + (ensure_pos_similarity ? 1u : 0u);
    if (n > 8192 || d1 > pg::kSsdMaxDim) {
        pg::set_error("pg_ssd: %u candidates x %u dims unsupported (<= 8192 x %u)", n, d1, pg::kSsdMaxDim);
        return PG_ERR_UNSUPPORTED;
    }
    for (uint32_t i = 0; i < n; ++i)
        PG_REQUIRE(cand_rows[i] < t->rows, "pg_ssd: candidate row %u outside table", cand_rows[i]);

    std::vector<double> quality(n);
    const bool bail = !pg::ssd_norm_quality_host(rel, n, norm_quality_score, quality.data());
    if (bail) {        // "all item score are zeros": the reference returns the items unchanged
        for (uint32_t i = 0; i < n; ++i) out_idx[i] = i;
        *out_count = n;
        if (out_quality) for (uint32_t i = 0; i < n; ++i) out_quality[i] = rel[i];
        return PG_OK;
    }
    if (out_quality) for (uint32_t i = 0; i < n; ++i) out_quality[i] = quality[i];

    const uint32_t T = n < topn ? n : topn;
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    void* io;
    int rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    if ((rc = pg::scratch_reserve(ctx, 5, al((size_t)n * 4) + al((size_t)n * 8) + al((size_t)T * 4), &io))) return rc;
    uint32_t* d_cand = (uint32_t*)io;
    double* d_rel = (double*)((char*)io + al((size_t)n * 4));
    uint32_t* d_out = (uint32_t*)((char*)io + al((size_t)n * 4) + al((size_t)n * 8));
    PG_HIP(hipMemcpyAsync(d_cand, cand_rows, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_rel, quality.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pg::ssd_run_locked(ctx, t, d_cand, d_rel, 1, n, gamma, T, window, normalize_emb, ensure_pos_similarity, use_ssd_star, d_out)))
        return rc;
    PG_HIP(hipMemcpyAsync(out_idx, d_out, (size_t)T * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    *out_count = T;
    return PG_OK;
}

}  // extern "C"
