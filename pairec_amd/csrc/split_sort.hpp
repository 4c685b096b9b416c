// split_sort.hpp — lists of up to 8192 keys sorted by MANY small workgroups each, for calls that carry fewer lists than the
// chip has CUs (per-request callers: 1 … ~128 requests per batch; sort/item_rank_score.go:26-32 and the final order of
// FaissModel.Run's top-K, one list per request either way).
//
// The one-workgroup network (sort_kernel_reg / final_kernel_reg) takes 56-75 us for a 5 000-item list whatever the chip is
// doing, and a 32-request batch leaves 224 CUs idle meanwhile; counting ranks (sort_rank_kernel / final_rank_kernel) spreads
// a list over the chip but is n^2 work: 31-63 us at eight lists.  Here:
//   1. split_sort_runs_kernel: every 512-slot piece of a list is sorted by ONE WAVE, in registers — the network of
//      bitonic_reg.hpp up to kk = 512 touches neither LDS nor a barrier — and written out as a sorted run;
//   2. split_sort_merge_kernel: one workgroup per piece again; it stages the list's runs in LDS (<= 64 KB) and an element's
//      final position is its position in its own run + for every other run the count a 10-step binary search returns:
//      keys below it — or, in runs of LOWER list positions when ties go by position, keys not above it.  n * (n / 512) * 10
//      LDS reads per list instead of n^2 compares.
// Same total order as the network: (key, position) ascending, callers complement keys for descending order.
// The Policy supplies the list (count, key of element i) and takes the result (store at rank; tail behind the last element).
#pragma once
#include "common.hpp"
#include "bitonic_reg.hpp"

namespace pg {

constexpr uint32_t kSplitRun = 512;                    // slots per run = one wave x kBitonicE
constexpr uint32_t kSplitMaxItems = 8192;              // the list's runs in LDS: 64 KB

// grid (pieces, lists), 64 threads
template <class Policy>
__global__ __launch_bounds__(64) void split_sort_runs_kernel(Policy pol, uint64_t* __restrict__ run_keys,
                                                             uint32_t* __restrict__ run_idx, uint32_t stride) {
    const uint32_t seg = blockIdx.y, base = blockIdx.x * kSplitRun, lane = threadIdx.x;
    const uint32_t n = pol.count(seg);
    if (base >= n) return;
    uint64_t k[kBitonicE];
    uint32_t ix[kBitonicE];
#pragma unroll
    for (int u = 0; u < kBitonicE; ++u) {
        const uint32_t i = base + lane * kBitonicE + u;
        k[u] = i < n ? pol.key(seg, i) : ~0ull;        // padding sorts last (behind a real ~0 key: its position is larger)
        ix[u] = i < n ? i : 0xFFFFFFFFu;
    }
    __shared__ char unused_lds[16];
    bitonic_sort_reg<Policy::kWithIdx, true>(k, ix, kSplitRun, *reinterpret_cast<BitonicLds*>(unused_lds), n - base);
    uint64_t* const ok = run_keys + (size_t)seg * stride + base + lane * kBitonicE;
    uint32_t* const oi = run_idx + (size_t)seg * stride + base + lane * kBitonicE;
#pragma unroll
    for (int u = 0; u < kBitonicE; ++u) {
        if (base + lane * kBitonicE + u < n) {         // (real elements lead the run)
            ok[u] = k[u];
            if (Policy::kWithIdx) oi[u] = ix[u];
        }
    }
}

// grid (pieces, lists), 256 threads, dynamic LDS = pieces * 512 * 8
template <class Policy>
__global__ __launch_bounds__(256) void split_sort_merge_kernel(Policy pol, const uint64_t* __restrict__ run_keys,
                                                               const uint32_t* __restrict__ run_idx, uint32_t stride) {
    extern __shared__ __attribute__((aligned(16))) uint64_t ss_keys[];
    const uint32_t seg = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = pol.count(seg);
    pol.tail(seg, n, part * 256u + tid, gridDim.x * 256u);
    const uint32_t base = part * kSplitRun;
    if (base >= n) return;
    const uint32_t parts = (n + kSplitRun - 1) / kSplitRun;
    const uint64_t* const in = run_keys + (size_t)seg * stride;
    // own positions first (their loads overlap the staging): e = tid, tid + 256
    uint32_t my_ix[2] = {0, 0};
    if (Policy::kWithIdx) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t i = base + tid + 256u * s;
            my_ix[s] = run_idx[(size_t)seg * stride + (i < n ? i : base)];
        }
    }
    for (uint32_t i0 = tid; i0 < n; i0 += 16u * 256u) {                    // sixteen independent loads per thread in flight
        uint64_t kv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t i = i0 + (uint32_t)u * 256u;
            kv[u] = in[i < n ? i : 0u];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t i = i0 + (uint32_t)u * 256u;
            if (i < n) ss_keys[i] = kv[u];
        }
    }
    __syncthreads();
    uint64_t mine[2];
    uint32_t rank[2];
    bool real[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t e = tid + 256u * s;
        real[s] = base + e < n;
        mine[s] = real[s] ? ss_keys[base + e] : 0ull;
        rank[s] = e;
    }
    // other runs, four at a time: eight searches in flight per thread
    constexpr int A = 4;
    for (uint32_t r0 = 0; r0 < parts; r0 += A) {
        uint32_t pos[A][2];
        uint32_t cnt[A];
        const uint64_t* rb[A];
        bool le[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const uint32_t r = r0 + a;
            const uint32_t rr = r < parts ? r : parts - 1;
            const uint32_t c = n - rr * kSplitRun;
            cnt[a] = (r < parts && r != part) ? (c < kSplitRun ? c : kSplitRun) : 0u;
            rb[a] = ss_keys + rr * kSplitRun;
            le[a] = Policy::kWithIdx && r < part;      // ties go by position: every key of an earlier run precedes an equal key here
            pos[a][0] = pos[a][1] = 0;
        }
#pragma unroll
        for (uint32_t st = kSplitRun; st >= 1; st >>= 1) {
#pragma unroll
            for (int a = 0; a < A; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint32_t j = pos[a][s] + st;
                    const bool in_run = j <= cnt[a];
                    const uint64_t x = rb[a][in_run ? j - 1 : 0u];
                    const bool before = le[a] ? (x <= mine[s]) : (x < mine[s]);
                    pos[a][s] = (in_run && before) ? j : pos[a][s];
                }
        }
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int s = 0; s < 2; ++s) rank[s] += pos[a][s];
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
        if (real[s]) pol.store(seg, rank[s], mine[s], my_ix[s]);
}

// host side: both launches on the context's stream; the runs live in scratch slot 7 (transient, as the other users of the slot)
template <class Policy>
int split_sort_launch(pg_ctx* ctx, const Policy& pol, uint32_t n_lists, uint32_t max_items) {
    const uint32_t parts = (max_items + kSplitRun - 1) / kSplitRun, stride = parts * kSplitRun;
    void* p;
    int rc;
    if ((rc = scratch_reserve(ctx, 7, (size_t)n_lists * stride * 12, &p))) return rc;
    uint64_t* const keys = (uint64_t*)p;
    uint32_t* const idx = (uint32_t*)(keys + (size_t)n_lists * stride);
    const size_t lds = (size_t)stride * 8;
    if ((rc = ensure_dyn_lds(ctx, (const void*)split_sort_merge_kernel<Policy>, lds))) return rc;
    split_sort_runs_kernel<Policy><<<dim3(parts, n_lists), 64, 0, ctx->stream>>>(pol, keys, idx, stride);
    split_sort_merge_kernel<Policy><<<dim3(parts, n_lists), 256, lds, ctx->stream>>>(pol, keys, idx, stride);
    PG_HIP(hipGetLastError());
    ctx->stats.sort_split_calls++;
    return PG_OK;
}

// Which sort a call takes (event-timed on MI355X, lists of 5 000: scripts/dev/sort_sweep.py): counting ranks is n^2 per list
// but one short launch — 18-21 us for one or two lists, 46 at four; the split sort 35 us up to 16 lists, 44 at 32, 62 at 96;
// the one-workgroup network 61-68 whatever the count (31 at 1 500 items, 70 at 8 192).  So: counting while lists x n^2 stays
// under rank_sort_work, else the split sort while the call holds at most ~450 K items in at most split_sort_max lists, else
// the network.  Lists beyond 8 192 items have neither alternative: counting up to kRankSortMaxSegments lists as before.
inline bool rank_sort_applies(const pg_ctx* ctx, uint32_t n_lists, uint32_t max_items, double work_scale = 1.0) {
    if (ctx->knobs.sort_lds || max_items > kRankSortMaxItems || n_lists > ctx->knobs.rank_sort_max) return false;
    if (max_items > kSplitMaxItems) return n_lists <= kRankSortMaxSegments;
    return (double)n_lists * max_items * max_items <= ctx->knobs.rank_sort_work * work_scale;
}
inline bool split_sort_applies(const pg_ctx* ctx, uint32_t n_lists, uint32_t max_items) {
    return max_items > 2 * kSplitRun && max_items <= kSplitMaxItems && n_lists <= ctx->knobs.split_sort_max &&
           (uint64_t)n_lists * max_items <= 450000u && !ctx->knobs.sort_lds;
}

}  // namespace pg
