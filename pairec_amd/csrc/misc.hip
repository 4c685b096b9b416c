// misc.hip — small glue kernels that keep a request batch on the device between stages.
#include "common.hpp"

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

}  // namespace pg

extern "C" {

int pg_rows_to_local_dev(pg_ctx* ctx, const pg_table* t, const uint64_t* d_rows, uint32_t n,
                         uint32_t* d_local, uint8_t* d_owned) {
    PG_REQUIRE(ctx && t && (n == 0 || (d_rows && d_local)), "pg_rows_to_local_dev: NULL argument");
    if (n == 0) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::rows_to_local_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(d_rows, n, t->row_offset, t->rows, d_local, d_owned);
    PG_HIP(hipGetLastError());
    return PG_OK;
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
