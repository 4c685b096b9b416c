// bitonic_reg.hpp — workgroup bitonic sort of up to 8192 keys with 8 elements per thread in registers.
//   partner inside the thread (j < 8)            → register compare-exchange, no memory traffic
//   partner in the same wave (8 <= j < 512)      → lane exchange, no barrier: DPP for lane distance 1, 2, 8 (VALU
//                                                   rate, no LDS crossbar), ds_swizzle for 4 and 16, ds_bpermute for 32
//   partner in another wave (j >= 512)           → LDS exchange ([u][thread] layout: conflict-free)
// 10 of the 91 sub-stages of an 8192-element sort touch LDS.  Ascending by (key, index); callers that
// want descending order complement their keys.  Position i = 8*t + u lives in thread t, slot u.
#pragma once
#include <cstdint>
#include <type_traits>
#include <hip/hip_runtime.h>

namespace pg {

constexpr int kBitonicE = 8;
constexpr uint32_t kBitonicMax = 8192;
constexpr uint32_t kRankSortMaxItems = 16384;     // ... of up to this many items each (the list's keys sit in LDS: 128 KB)
constexpr uint32_t kRankSortMaxSegments = 8;      // (default of Knobs::rank_sort_max)    // up to this many lists per call: ranks by counting, spread over the chip (final_rank_kernel, sort_rank_kernel)

// value of lane (l ^ M) — M a compile-time power of two below 64
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
    else if constexpr (M == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);              // xor 4 (bit mode)
    else if constexpr (M == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);             // xor 16
    else return (uint32_t)__shfl_xor((int)v, M, 64);
}

// the wave's maximum of x with NaN passed over (v_max_f64), as a wave-uniform value (scalar registers): six lane exchanges —
// the VALUE half of a "first maximum, NaN skipped" argmax (floats.MaxIdx); the index then comes from a ballot of x == max
template <int K0, int K1, class F>
__device__ __forceinline__ void lane_static_for(F&& f) {
    if constexpr (K0 < K1) {
        f(std::integral_constant<int, K0>{});
        lane_static_for<K0 + 1, K1>(f);
    }
}
__device__ __forceinline__ double uniform_f64(double x) {
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double wave_max_f64(double m) {
    lane_static_for<0, 6>([&](auto sc) {
        constexpr int off = 1 << decltype(sc)::value;
        const uint64_t bb = (uint64_t)__double_as_longlong(m);
        const uint32_t lo = lane_xor<off>((uint32_t)bb), hi = lane_xor<off>((uint32_t)(bb >> 32));
        m = fmax(m, __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)));
    });
    return uniform_f64(m);
}

struct BitonicLds {
    uint64_t xk[kBitonicE][1024];
    uint32_t xi[kBitonicE][1024];
};

// P: power of two, 512 <= P <= 8192; threads t < P/8 are active (whole waves), every thread of the
// 1024-thread workgroup must call (barriers).  WITH_IDX = false ignores ix (keys are unique).
// n_real: elements at positions >= n_real are padding — all equal and not below any real key.  In phase kk elements stay
// inside their aligned block of kk, so a block that starts at or behind n_real holds padding only and has nothing to
// sort: a wave whose 512 elements lie in such blocks skips the phase's compare-exchanges (not its barriers).  5 000 of
// 8 192: six of sixteen waves idle through the 55 stages up to kk = 1024, four through the 11 of kk = 2048 — the network is
// VALU-bound, four waves per SIMD.
// WAVE_ONLY: P = 512 in a one-wave workgroup — the branch that crosses waves is not compiled, `lds` is never touched.
template <bool WITH_IDX, bool WAVE_ONLY = false>
__device__ __forceinline__ void bitonic_sort_reg(uint64_t (&k)[kBitonicE], uint32_t (&ix)[kBitonicE], uint32_t P,
                                                 BitonicLds& lds, uint32_t n_real = 0xFFFFFFFFu) {
    const uint32_t t = threadIdx.x;
    const bool act_all = t < P / kBitonicE;
    const uint32_t wbase = (t & ~63u) * kBitonicE;        // this wave's first element
    auto lt = [](uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
        return WITH_IDX ? (ka < kb || (ka == kb && ia < ib)) : (ka < kb);
    };
    for (uint32_t kk = 2; kk <= P; kk <<= 1) {
        const bool act = act_all && (kk >= 64u * kBitonicE ? (wbase & ~(kk - 1)) : wbase) < n_real;    // (wave-uniform)
        // ---- partners in other threads (j >= 8, so kk >= 16): direction is per thread
        const bool dir = (t & (kk / kBitonicE)) == 0;     // this thread's sub-sequence ascends
        for (uint32_t j = kk >> 1; j >= (uint32_t)kBitonicE; j >>= 1) {
            const uint32_t m = j / kBitonicE;             // partner thread = t ^ m
            const bool keep_min = ((t & m) == 0) == dir;
            if (!WAVE_ONLY && m >= 64) {
                if (act) {
#pragma unroll
                    for (int u = 0; u < kBitonicE; ++u) {
                        lds.xk[u][t] = k[u];
                        if (WITH_IDX) lds.xi[u][t] = ix[u];
                    }
                }
                __syncthreads();
                if (act) {
#pragma unroll
                    for (int u = 0; u < kBitonicE; ++u) {
                        const uint64_t ok = lds.xk[u][t ^ m];
                        const uint32_t oi = WITH_IDX ? lds.xi[u][t ^ m] : 0u;
                        const bool mine_first = lt(k[u], ix[u], ok, oi);
                        if (mine_first != keep_min) { k[u] = ok; ix[u] = oi; }
                    }
                }
                __syncthreads();
            } else if (act) {
                auto exch = [&](auto m_tag) {
                    constexpr int M = decltype(m_tag)::value;
#pragma unroll
                    for (int u = 0; u < kBitonicE; ++u) {
                        const uint32_t olo = lane_xor<M>((uint32_t)k[u]);
                        const uint32_t ohi = lane_xor<M>((uint32_t)(k[u] >> 32));
                        const uint32_t oi = WITH_IDX ? lane_xor<M>(ix[u]) : 0u;
                        const uint64_t ok = ((uint64_t)ohi << 32) | olo;
                        const bool mine_first = lt(k[u], ix[u], ok, oi);
                        if (mine_first != keep_min) { k[u] = ok; ix[u] = oi; }
                    }
                };
                switch (m) {
                    case 1: exch(std::integral_constant<int, 1>{}); break;
                    case 2: exch(std::integral_constant<int, 2>{}); break;
                    case 4: exch(std::integral_constant<int, 4>{}); break;
                    case 8: exch(std::integral_constant<int, 8>{}); break;
                    case 16: exch(std::integral_constant<int, 16>{}); break;
                    default: exch(std::integral_constant<int, 32>{}); break;
                }
            }
        }
        // ---- partners inside the thread (j = min(kk/2, 4) .. 1)
        if (act) {
#pragma unroll
            for (int jj = kBitonicE / 2; jj >= 1; jj >>= 1) {
                if ((uint32_t)jj >= kk) continue;
#pragma unroll
                for (int u = 0; u < kBitonicE; ++u) {
                    if (u & jj) continue;
                    const int l = u | jj;
                    const bool up = ((t * kBitonicE + (uint32_t)u) & kk) == 0;      // ascending sub-sequence
                    const bool first = lt(k[u], ix[u], k[l], ix[l]);
                    if (first != up) {
                        const uint64_t tk = k[u]; k[u] = k[l]; k[l] = tk;
                        const uint32_t ti = ix[u]; ix[u] = ix[l]; ix[l] = ti;
                    }
                }
            }
        }
    }
}

}  // namespace pg
