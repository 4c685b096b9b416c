// misc.hip — small glue kernels that keep a request batch on the device between stages.
#include "pipeline.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace pg {

__global__ void rows_to_local_kernel(const uint64_t* __restrict__ rows, uint32_t n, uint64_t off,
                                     uint64_t nrows, uint32_t* __restrict__ local,
                                     uint8_t* __restrict__ owned) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = rows[i];
    const bool mine = r != ~0ull && r >= off && r - off < nrows;
    local[i] = mine ? (uint32_t)(r - off) : 0u;
    if (owned) owned[i] = mine ? 1 : 0;
}

__global__ void widen_kernel(const float* __restrict__ in, uint32_t n, double* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

// read-only streaming probe: the measured HBM ceiling that bench.py reports beside the nominal 8 TB/s
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hbm_probe_kernel(const u32x4* __restrict__ p, uint64_t n16,
                                                        uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(&p[i]);
        const u32x4 b = __builtin_nontemporal_load(&p[i + stride]);
        const u32x4 c = __builtin_nontemporal_load(&p[i + 2 * stride]);
        const u32x4 d = __builtin_nontemporal_load(&p[i + 3 * stride]);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) {
        const u32x4 a = __builtin_nontemporal_load(&p[i]);
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9E3779B9u) *sink = acc;      // never true in practice; keeps the loads alive
}

// req_offsets[r] = r * k (uniform candidate count per request)
__global__ void uniform_offsets_kernel(uint32_t nq, uint32_t k, uint32_t* __restrict__ off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nq) off[i] = i * k;
}

int rows_to_local_locked(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t n, uint32_t* d_local,
                         uint8_t* d_owned) {
    if (n == 0) return PG_OK;
    rows_to_local_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(d_rows, n, t->row_offset, t->rows, d_local, d_owned);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

int uniform_offsets_locked(pg_ctx* ctx, uint32_t nq, uint32_t k, uint32_t* d_off) {
    uniform_offsets_kernel<<<(nq + 256) / 256, 256, 0, ctx->stream>>>(nq, k, d_off);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // namespace pg

extern "C" {

int pg_hbm_read_probe(pg_ctx* ctx, const pg_table* t, int reps, double* out_gbps) {
    PG_REQUIRE(ctx && t && out_gbps && reps > 0, "pg_hbm_read_probe: bad argument");
    const void* d_buf = t->d;
    const uint64_t bytes = t->rows * (uint64_t)t->dim * 4;
    PG_REQUIRE(bytes >= (1u << 20), "pg_hbm_read_probe: table smaller than 1 MiB");
    std::lock_guard<std::mutex> g(ctx->mu);
    void* sink;
    int rc;
    if ((rc = pg::scratch_reserve(ctx, 4, 4096, &sink))) return rc;
    const uint64_t n16 = bytes / 16;
    const int grid = ctx->num_cus * 8;
    double best = 0.0;
    for (int r = 0; r < reps + 1; ++r) {      // first launch is a warm-up
        PG_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
        pg::hbm_probe_kernel<<<grid, 256, 0, ctx->stream>>>((const pg::u32x4*)d_buf, n16, (uint32_t*)sink + 900);
        PG_HIP(hipGetLastError());
        PG_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
        PG_HIP(hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        PG_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        const double gbps = (double)(n16 * 16) / (ms * 1e-3) / 1e9;
        if (r > 0 && gbps > best) best = gbps;
    }
    *out_gbps = best;
    return PG_OK;
}

int pg_rows_to_local_dev(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t n,
                         uint32_t* d_local, uint8_t* d_owned) {
    PG_REQUIRE(ctx && t && (n == 0 || (d_rows && d_local)), "pg_rows_to_local_dev: NULL argument");
    if (n == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::rows_to_local_locked(ctx, t, d_rows, n, d_local, d_owned);
}

int pg_widen_f32_dev(pg_ctx* ctx, const float* d_in, uint32_t n, double* d_out) {
    PG_REQUIRE(ctx && (n == 0 || (d_in && d_out)), "pg_widen_f32_dev: NULL argument");
    if (n == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::widen_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(d_in, n, d_out);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // extern "C"
