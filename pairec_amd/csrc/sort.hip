// sort.hip — segmented score sort on the device.
//
// Replaces ItemRankScoreSort (descending; sort/item_rank_score.go:26-32, the default sort,
// sort/sort.go:103-107) and ItemScoreSort (ascending; sort/item_score.go:36-41); both order
// []*module.Item by the float64 Item.Score with Go's unstable sort.Sort.  Here the order is total:
// score (±0 equal, NaN last in both directions), then input index ascending.
// One workgroup per request; bitonic network over (ordered-u64 key, u32 index) pairs held in LDS
// for segments up to 8192 items, in a global scratch slab beyond that.
#include "common.hpp"
#include "bitonic_reg.hpp"
#include "split_sort.hpp"

namespace pg {

__device__ __forceinline__ uint64_t f64_ordered_bits(double d) {
    if (d == 0.0) d = 0.0;                        // -0 → +0 (Go's < treats them equal)
    const uint64_t b = (uint64_t)__double_as_longlong(d);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// "a sorts before b"
__device__ __forceinline__ bool before(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib, bool desc) {
    if (ka != kb) return desc ? (ka > kb) : (ka < kb);
    return ia < ib;
}

constexpr uint32_t kSortLdsMax = 8192;

__global__ __launch_bounds__(1024) void sort_kernel(const double* __restrict__ scores,
                                                    const uint32_t* __restrict__ seg_offsets,
                                                    int desc, uint64_t* __restrict__ g_keys,
                                                    uint32_t* __restrict__ g_idx, uint32_t g_stride,
                                                    uint32_t* __restrict__ out_order) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const uint32_t seg = blockIdx.x, tid = threadIdx.x;
    const uint32_t b = seg_offsets[seg], e = seg_offsets[seg + 1];
    const uint32_t n = e - b;
    if (n == 0) return;
    uint32_t P = 2;
    while (P < n) P <<= 1;
    uint64_t* keys;
    uint32_t* idx;
    if (P <= kSortLdsMax) {
        keys = reinterpret_cast<uint64_t*>(smem_raw);
        idx = reinterpret_cast<uint32_t*>(smem_raw + (size_t)kSortLdsMax * 8);
    } else {
        keys = g_keys + (size_t)seg * g_stride;
        idx = g_idx + (size_t)seg * g_stride;
    }
    const uint64_t pad_key = desc ? 0ull : ~0ull;       // NaN and padding sort last
    for (uint32_t i = tid; i < P; i += 1024) {
        if (i < n) {
            const double s = scores[b + i];
            keys[i] = (s != s) ? pad_key : f64_ordered_bits(s);
            idx[i] = i;
        } else {
            keys[i] = pad_key;
            idx[i] = 0xFFFFFFFFu;
        }
    }
    __syncthreads();
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < P; i += 1024) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t ka = keys[i], kb = keys[ixj];
                    const uint32_t ia = idx[i], ib = idx[ixj];
                    const bool fwd = (i & k) == 0;          // this sub-sequence sorts "before"-first
                    const bool a_first = before(ka, ia, kb, ib, desc != 0);
                    if (a_first != fwd) {
                        keys[i] = kb; keys[ixj] = ka;
                        idx[i] = ib; idx[ixj] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < n; i += 1024) out_order[b + i] = idx[i];
}

// Segments of up to 8192 items: the register-resident bitonic network of bitonic_reg.hpp (10 of the 91
// sub-stages of an 8192-element sort touch LDS; the plain LDS network above: all 91, with a barrier each).
// Descending order is the ascending network on complemented keys; ties by index either way.
__global__ __launch_bounds__(1024) void sort_kernel_reg(const double* __restrict__ scores,
                                                        const uint32_t* __restrict__ seg_offsets, int desc,
                                                        uint32_t* __restrict__ out_order) {
    __shared__ BitonicLds lds;
    const uint32_t seg = blockIdx.x, t = threadIdx.x;
    const uint32_t b = seg_offsets[seg], e = seg_offsets[seg + 1];
    const uint32_t n = e - b;
    if (n == 0) return;
    uint32_t P = 512;                                  // at least one full wave of 8-element threads
    while (P < n) P <<= 1;
    const bool act = t < P / kBitonicE;
    uint64_t k[kBitonicE];
    uint32_t ix[kBitonicE];
#pragma unroll
    for (int u = 0; u < kBitonicE; ++u) {
        const uint32_t i = t * kBitonicE + u;
        if (act && i < n) {
            const double sc = scores[b + i];
            const uint64_t key = (sc != sc) ? (desc ? 0ull : ~0ull) : f64_ordered_bits(sc);   // NaN sorts last
            k[u] = desc ? ~key : key;
            ix[u] = i;
        } else {
            k[u] = ~0ull;                               // padding sorts last
            ix[u] = 0xFFFFFFFFu;
        }
    }
    bitonic_sort_reg<true>(k, ix, P, lds, n);
    if (act) {
#pragma unroll
        for (int u = 0; u < kBitonicE; ++u) {
            const uint32_t i = t * kBitonicE + u;
            if (i < n) out_order[b + i] = ix[u];
        }
    }
}

// A handful of segments of up to 8192 items: ranks by counting, 64 items per workgroup (see final_rank_kernel in
// recall.hip — the network above keeps one CU busy for 66 us per segment while the rest of the chip idles).
// item i's position = #{j : key_j < key_i} + #{j < i : key_j = key_i}: the same order as the network (ties by index).
// EPB items per workgroup, 256 / EPB threads per item (each counts within its share of the keys): 16 for one or two
// segments (a 5 000-item segment then spreads over 313 workgroups), 64 beyond.
template <int EPB>
__global__ __launch_bounds__(256) void sort_rank_kernel(const double* __restrict__ scores, const uint32_t* __restrict__ seg_offsets,
                                                        int desc, uint32_t* __restrict__ out_order) {
    constexpr uint32_t PARTS = 256 / EPB;
    extern __shared__ __attribute__((aligned(16))) uint64_t rk_keys[];       // [n rounded up to 32]
    __shared__ uint32_t part[PARTS][EPB];
    const uint32_t seg = blockIdx.y, tid = threadIdx.x;
    const uint32_t b = seg_offsets[seg], n = seg_offsets[seg + 1] - b;
    if (blockIdx.x * (uint32_t)EPB >= n) return;
    const uint32_t n8 = (n + 31u) & ~31u;
    // (eight independent loads per thread in flight: written as a plain strided loop this staging was 20 dependent
    // L2 round trips — most of the kernel)
    for (uint32_t i0 = tid; i0 < n8; i0 += 8u * 256u) {
        double sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + (uint32_t)u * 256u;
            sc[u] = scores[b + (i < n ? i : 0u)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = i0 + (uint32_t)u * 256u;
            if (i >= n8) break;
            uint64_t k = ~0ull;                                              // padding sorts last
            if (i < n) {
                const uint64_t key = (sc[u] != sc[u]) ? (desc ? 0ull : ~0ull) : f64_ordered_bits(sc[u]);   // NaN sorts last
                k = desc ? ~key : key;
            }
            rk_keys[i] = k;
        }
    }
    __syncthreads();
    const uint32_t el = tid % (uint32_t)EPB, p = tid / (uint32_t)EPB;
    const uint32_t e = blockIdx.x * (uint32_t)EPB + el;
    const uint64_t mine = e < n ? rk_keys[e] : 0ull;
    const uint32_t chunk = n8 / PARTS;                                     // n8 % 32 == 0: even chunks
    uint32_t j0 = p * chunk, j1 = j0 + chunk;
    // (the padding keys ~0 are never counted: nothing is above them, and an equal key — a NaN item — has j >= n > e.
    // Unrolled: one wave per SIMD here, so the LDS latency of a broadcast read is hidden only by the reads behind it)
    uint32_t below = 0;
#pragma unroll 4
    for (uint32_t j = j0; j < j1; j += 2) {
        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&rk_keys[j]);
        below += ((kk.x < mine) | ((kk.x == mine) & (j < e))) ? 1u : 0u;
        below += ((kk.y < mine) | ((kk.y == mine) & (j + 1 < e))) ? 1u : 0u;
    }
    part[p][el] = below;
    __syncthreads();
    if (p == 0 && e < n) {
        uint32_t r = 0;
#pragma unroll
        for (uint32_t i = 0; i < PARTS; ++i) r += part[i][el];
        out_order[b + r] = e;
    }
}

// the score sort as a split sort (split_sort.hpp): keys as in sort_kernel_reg, ties by input position
struct ScoreSortPolicy {
    static constexpr bool kWithIdx = true;
    const double* scores;
    const uint32_t* seg;
    int desc;
    uint32_t* out;
    __device__ uint32_t count(uint32_t s) const { return seg[s + 1] - seg[s]; }
    __device__ uint64_t key(uint32_t s, uint32_t i) const {
        const double sc = scores[seg[s] + i];
        const uint64_t key = (sc != sc) ? (desc ? 0ull : ~0ull) : f64_ordered_bits(sc);   // NaN sorts last
        return desc ? ~key : key;
    }
    __device__ void store(uint32_t s, uint32_t rank, uint64_t, uint32_t idx) const { out[seg[s] + rank] = idx; }
    __device__ void tail(uint32_t, uint32_t, uint32_t, uint32_t) const {}
};

int sort_dev_locked(pg_ctx* ctx, const double* d_scores, const uint32_t* d_seg, uint32_t n_seg,
                           uint32_t n_items, uint32_t max_seg, int desc, uint32_t* d_out) {
    if (n_seg == 0 || n_items == 0) return PG_OK;
    constexpr size_t lds = (size_t)kSortLdsMax * 12;
    int rc_attr;
    if ((rc_attr = ensure_dyn_lds(ctx, (const void*)sort_kernel, lds))) return rc_attr;
    uint64_t* g_keys = nullptr;
    uint32_t* g_idx = nullptr;
    uint32_t stride = 0;
    if (max_seg > kSortLdsMax) {
        stride = 2;
        while (stride < max_seg) stride <<= 1;
        void* p;
        int rc;
        if ((rc = scratch_reserve(ctx, 7, (size_t)n_seg * stride * 12, &p))) return rc;
        g_keys = (uint64_t*)p;
        g_idx = (uint32_t*)(g_keys + (size_t)n_seg * stride);
    }
    if (rank_sort_applies(ctx, n_seg, max_seg)) {
        const size_t rl = (size_t)((max_seg + 31u) & ~31u) * 8;
        if (n_seg <= 2) {
            if ((rc_attr = ensure_dyn_lds(ctx, (const void*)sort_rank_kernel<16>, rl))) return rc_attr;
            sort_rank_kernel<16><<<dim3((max_seg + 15) / 16, n_seg), 256, rl, ctx->stream>>>(d_scores, d_seg, desc, d_out);
        } else {
            if ((rc_attr = ensure_dyn_lds(ctx, (const void*)sort_rank_kernel<64>, rl))) return rc_attr;
            sort_rank_kernel<64><<<dim3((max_seg + 63) / 64, n_seg), 256, rl, ctx->stream>>>(d_scores, d_seg, desc, d_out);
        }
    } else if (split_sort_applies(ctx, n_seg, max_seg)) {
        int rc;
        if ((rc = split_sort_launch(ctx, ScoreSortPolicy{d_scores, d_seg, desc, d_out}, n_seg, max_seg))) return rc;
    } else if (max_seg <= kSortLdsMax && !ctx->knobs.sort_lds)
        sort_kernel_reg<<<n_seg, 1024, 0, ctx->stream>>>(d_scores, d_seg, desc, d_out);
    else
        sort_kernel<<<n_seg, 1024, lds, ctx->stream>>>(d_scores, d_seg, desc, g_keys, g_idx, stride, d_out);
    PG_HIP(hipGetLastError());
    ctx->stats.sort_calls++;
    ctx->stats.sort_items += n_items;
    return PG_OK;
}

}  // namespace pg

extern "C" {

int pg_sort_scores_dev(pg_ctx* ctx, const double* d_scores, const uint32_t* d_seg_offsets, uint32_t n_seg,
                       uint32_t n_items, uint32_t max_segment, int descending, uint32_t* d_out_order) {
    PG_REQUIRE(ctx && d_scores && d_seg_offsets && d_out_order, "pg_sort_scores_dev: NULL argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    // segment sizes live on the device: the caller's bound (or the worst case) sizes the scratch
    const uint32_t bound = (max_segment == 0 || max_segment > n_items) ? n_items : max_segment;
    return pg::sort_dev_locked(ctx, d_scores, d_seg_offsets, n_seg, n_items, bound, descending, d_out_order);
}

int pg_sort_scores(pg_ctx* ctx, const double* scores, const uint32_t* seg_offsets, uint32_t n_seg,
                   int descending, uint32_t* out_order) {
    PG_REQUIRE(ctx && seg_offsets, "pg_sort_scores: NULL argument");
    if (n_seg == 0) return PG_OK;
    PG_REQUIRE(seg_offsets[0] == 0, "pg_sort_scores: seg_offsets[0] must be 0");
    uint32_t max_seg = 0;
    for (uint32_t s = 0; s < n_seg; ++s) {
        PG_REQUIRE(seg_offsets[s + 1] >= seg_offsets[s], "pg_sort_scores: seg_offsets not monotone at %u", s);
        max_seg = std::max(max_seg, seg_offsets[s + 1] - seg_offsets[s]);
    }
    const uint32_t n = seg_offsets[n_seg];
    if (n == 0) return PG_OK;
    PG_REQUIRE(scores && out_order, "pg_sort_scores: NULL argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    void* buf;
    int rc;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    if ((rc = pg::scratch_reserve(ctx, 5, al((size_t)n * 8) + al((size_t)(n_seg + 1) * 4) + al((size_t)n * 4), &buf)))
        return rc;
    double* d_s = (double*)buf;
    uint32_t* d_o = (uint32_t*)((char*)buf + al((size_t)n * 8));
    uint32_t* d_r = (uint32_t*)((char*)d_o + al((size_t)(n_seg + 1) * 4));
    PG_HIP(hipMemcpyAsync(d_s, scores, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipMemcpyAsync(d_o, seg_offsets, (size_t)(n_seg + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
    if ((rc = pg::sort_dev_locked(ctx, d_s, d_o, n_seg, n, max_seg, descending, d_r))) return rc;
    PG_HIP(hipEventRecord(ctx->ev[5], ctx->stream));
    PG_HIP(hipMemcpyAsync(out_order, d_r, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    PG_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
    ctx->stats.last_sort_ms = ms;
    return PG_OK;
}

}  // extern "C"
