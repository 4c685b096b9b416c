// table.hip — HBM-resident embedding tables: the device-side replacement for pairec's
// module.VectorDao back-ends (module/vector_dao.go:13-15; redis/hologres/mysql/clickhouse/hbase/be
// implementations fetch one embedding per network round trip).  Rows are fp32, row-major,
// [rows][dim], 512 B per row at dim=128, so a recall scan streams them as full 128-B lines.
#include "common.hpp"
#include "synth.hpp"

namespace pg {

// One block = 256 rows.  Phase 1: thread-per-row computes the row's squared norm as the
// specification's k-ascending fmaf chain (must be sequential to be bit-reproducible).
// Phase 2: the block rewrites the same rows with coalesced 16-B stores.
template <int DIM>
__global__ __launch_bounds__(256) void table_fill_synth_kernel(float* __restrict__ out,
                                                              uint64_t rows, uint64_t row_offset,
                                                              uint64_t seed, int normalize) {
    __shared__ float inv_s[256];
    const uint64_t r0 = (uint64_t)blockIdx.x * 256;
    const uint64_t r = r0 + threadIdx.x;
    float inv = 1.0f;
    if (normalize && r < rows) {
        const uint64_t g = row_offset + r;
        float ss = 0.0f;
        for (int c = 0; c < DIM; ++c) {
            const float v = synth_value(seed, g, c, DIM);
            ss = __fmaf_rn(v, v, ss);
        }
        inv = 1.0f / sqrtf(ss);   // IEEE-rounded sqrt and divide (hipcc default for fp32)
    }
    inv_s[threadIdx.x] = inv;
    __syncthreads();
    constexpr int QPR = DIM / 4;                  // quads per row
    constexpr int RPI = 256 / QPR;                // rows per iteration
    const int q = threadIdx.x % QPR;
    for (int rr = threadIdx.x / QPR; rr < 256; rr += RPI) {
        const uint64_t row = r0 + rr;
        if (row >= rows) break;
        const uint64_t g = row_offset + row;
        const float s = inv_s[rr];
        float4 v;
        v.x = synth_value(seed, g, 4 * q + 0, DIM) * s;
        v.y = synth_value(seed, g, 4 * q + 1, DIM) * s;
        v.z = synth_value(seed, g, 4 * q + 2, DIM) * s;
        v.w = synth_value(seed, g, 4 * q + 3, DIM) * s;
        *reinterpret_cast<float4*>(out + row * DIM + 4 * q) = v;
    }
}

// Clustered rows (the third benchmark distribution: trained embeddings cluster, service/recall/hologres_vector_recall.go:23 runs
// against such tables): row g belongs to centre c = splitmix64((seed + 2) ^ g) mod n_centres; the centre is SURVEY.md 8d's
// normalised synthetic row c of seed + 1; x = centre + noise_scale * u, u uniform in [-1, 1) from seed + 3 (noise_scale =
// sigma sqrt(3 / dim): the noise vector's norm is ~ sigma, the centre's is 1); the row is then normalised like the synthetic
// ones.  No transcendental: oracle/oracle.c (orc_synth_mixture_rows) regenerates any slice bit for bit.
template <int DIM>
__global__ __launch_bounds__(256) void table_fill_mixture_kernel(float* __restrict__ out, uint64_t rows, uint64_t row_offset,
                                                                uint64_t seed, uint32_t n_centres, float noise_scale) {
    __shared__ float inv_c[256], inv_x[256];
    __shared__ uint32_t cen[256];
    const uint64_t r0 = (uint64_t)blockIdx.x * 256;
    const uint64_t r = r0 + threadIdx.x;
    if (r < rows) {
        const uint64_t g = row_offset + r;
        const uint32_t c = (uint32_t)(splitmix64((seed + 2) ^ g) % n_centres);
        float ss = 0.0f;
        for (int k = 0; k < DIM; ++k) {
            const float v = synth_value(seed + 1, c, k, DIM);
            ss = __fmaf_rn(v, v, ss);
        }
        const float ic = 1.0f / sqrtf(ss);
        float sx = 0.0f;
        for (int k = 0; k < DIM; ++k) {
            const float x = __fmaf_rn(synth_value(seed + 3, g, k, DIM), noise_scale, synth_value(seed + 1, c, k, DIM) * ic);
            sx = __fmaf_rn(x, x, sx);
        }
        cen[threadIdx.x] = c;
        inv_c[threadIdx.x] = ic;
        inv_x[threadIdx.x] = 1.0f / sqrtf(sx);
    }
    __syncthreads();
    constexpr int QPR = DIM / 4, RPI = 256 / QPR;
    const int q = threadIdx.x % QPR;
    for (int rr = threadIdx.x / QPR; rr < 256; rr += RPI) {
        const uint64_t row = r0 + rr;
        if (row >= rows) break;
        const uint64_t g = row_offset + row;
        const uint32_t c = cen[rr];
        const float ic = inv_c[rr], ix = inv_x[rr];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = __fmaf_rn(synth_value(seed + 3, g, 4 * q + e, DIM), noise_scale, synth_value(seed + 1, c, 4 * q + e, DIM) * ic) * ix;
        *reinterpret_cast<float4*>(out + row * DIM + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Gaussian rows for the second benchmark distribution (real embeddings look Gaussian, not uniform): Box-Muller on
// one splitmix64 draw per element, z = sqrt(-2 ln u1) cos(2 pi u2) with u1 from the draw's upper 53 bits (tails to
// 8.5 sigma) evaluated in fp64, value = (float)(z * scale).  Device-defined: tests compare against the rows they
// download, not against a CPU regeneration (log / cos differ by an ulp between libms).
__global__ __launch_bounds__(256) void table_fill_gauss_kernel(float* __restrict__ out, uint64_t n_elems,
                                                              uint64_t elem_offset, uint64_t seed, double scale) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += stride) {
        const uint64_t u = splitmix64(seed ^ (elem_offset + i));
        const double u1 = (double)((u >> 11) + 1) * (1.0 / 9007199254740992.0);       // (0, 1]
        const double u2 = (double)(splitmix64(u) >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
        out[i] = (float)(sqrt(-2.0 * log(u1)) * cospi(2.0 * u2) * scale);
    }
}

// one wave-half (32 lanes x 16 B = 512 B) per gathered row at dim=128
__global__ void table_gather_kernel(const float* __restrict__ tab, uint32_t dim,
                                    const uint32_t* __restrict__ rows, uint32_t n,
                                    float* __restrict__ out) {
    const uint32_t qpr = dim / 4;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t i = gid / qpr;
    const uint32_t q = gid % qpr;
    if (i >= n) return;
    const float4 v = *reinterpret_cast<const float4*>(tab + (uint64_t)rows[i] * dim + 4 * q);
    *reinterpret_cast<float4*>(out + i * dim + 4 * q) = v;
}

int table_gather_locked(pg_ctx* ctx, const pg_table* t, const uint32_t* d_rows, uint32_t n, float* d_out) {
    if (n == 0) return PG_OK;
    const uint64_t threads = (uint64_t)n * (t->dim / 4);
    table_gather_kernel<<<(uint32_t)((threads + 255) / 256), 256, 0, ctx->stream>>>(t->d, t->dim, d_rows, n, d_out);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // namespace pg

extern "C" {

int pg_table_create(pg_ctx* ctx, uint64_t rows, uint32_t dim, uint64_t row_offset, pg_table** out) {
    PG_REQUIRE(ctx && out, "pg_table_create: NULL argument");
    PG_REQUIRE(dim >= 64 && dim % 64 == 0 && dim <= 256,
               "pg_table_create: dim=%u unsupported (multiple of 64, <= 256)", dim);
    PG_REQUIRE(rows > 0 && rows < (1ull << 32), "pg_table_create: rows=%llu must be in [1, 2^32)",
               (unsigned long long)rows);
    PG_REQUIRE(row_offset + rows < (1ull << 32),
               "pg_table_create: global row ids must fit in 32 bits (offset %llu + rows %llu)",
               (unsigned long long)row_offset, (unsigned long long)rows);
    std::lock_guard<std::mutex> g(ctx->mu);
    PG_HIP(hipSetDevice(ctx->device));
    pg_table* t = new pg_table();
    t->rows = rows;
    t->dim = dim;
    t->row_offset = row_offset;
    // +64 rows of slack so tile loaders may read (never use) a few rows past the end
    hipError_t e = hipMalloc((void**)&t->d, (rows + 64) * (size_t)dim * sizeof(float));
    if (e != hipSuccess) {
        pg::set_error("pg_table_create: hipMalloc(%.1f GB) failed: %s",
                      (double)(rows * dim * 4) / 1e9, hipGetErrorString(e));
        delete t;
        return PG_ERR_NOMEM;
    }
    PG_HIP(hipMemsetAsync(t->d + rows * (size_t)dim, 0, 64 * (size_t)dim * sizeof(float), ctx->stream));
    *out = t;
    return PG_OK;
}

int pg_table_destroy(pg_ctx* ctx, pg_table* t) {
    PG_REQUIRE(ctx, "pg_table_destroy: ctx is NULL");
    if (!t) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    PG_HIP(hipStreamSynchronize(ctx->stream));
    if (t->d) PG_HIP(hipFree(t->d));
    if (t->d16) PG_HIP(hipFree(t->d16));
    if (t->d8) PG_HIP(hipFree(t->d8));
    if (t->d8r) PG_HIP(hipFree(t->d8r));
    if (t->d4) PG_HIP(hipFree(t->d4));
    if (t->d4s) PG_HIP(hipFree(t->d4s));
    if (t->dnorm2) PG_HIP(hipFree(t->dnorm2));
    if (t->d_pred) PG_HIP(hipFree(t->d_pred));
    if (t->d_nx) PG_HIP(hipFree(t->d_nx));
    if (t->d_nxmin) PG_HIP(hipFree(t->d_nxmin));
    if (t->d_row_map) PG_HIP(hipFree(t->d_row_map));
    delete t;
    return PG_OK;
}

int pg_table_fill_synthetic(pg_ctx* ctx, pg_table* t, uint64_t seed, int normalize) {
    PG_REQUIRE(ctx && t, "pg_table_fill_synthetic: NULL argument");
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableWrite w(t);
    const uint32_t blocks = (uint32_t)((t->rows + 255) / 256);
    switch (t->dim) {
        case 64:
            pg::table_fill_synth_kernel<64><<<blocks, 256, 0, ctx->stream>>>(t->d, t->rows, t->row_offset, seed, normalize);
            break;
        case 128:
            pg::table_fill_synth_kernel<128><<<blocks, 256, 0, ctx->stream>>>(t->d, t->rows, t->row_offset, seed, normalize);
            break;
        case 192:
            pg::table_fill_synth_kernel<192><<<blocks, 256, 0, ctx->stream>>>(t->d, t->rows, t->row_offset, seed, normalize);
            break;
        case 256:
            pg::table_fill_synth_kernel<256><<<blocks, 256, 0, ctx->stream>>>(t->d, t->rows, t->row_offset, seed, normalize);
            break;
        default:
            pg::set_error("pg_table_fill_synthetic: dim=%u unsupported", t->dim);
            return PG_ERR_UNSUPPORTED;
    }
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(ctx->stream));
    t->stats_valid = false;
    t->screen_overflow_streak = t->screen_backoff = 0;
    t->nx_valid = false;
    return PG_OK;
}

int pg_table_fill_gaussian(pg_ctx* ctx, pg_table* t, uint64_t seed, float sigma) {
    PG_REQUIRE(ctx && t, "pg_table_fill_gaussian: NULL argument");
    PG_REQUIRE(sigma > 0.0f && sigma < 1e30f, "pg_table_fill_gaussian: sigma must be positive and finite");
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableWrite w(t);
    const uint64_t n = t->rows * (uint64_t)t->dim;
    pg::table_fill_gauss_kernel<<<(uint32_t)ctx->num_cus * 16, 256, 0, ctx->stream>>>(t->d, n, t->row_offset * (uint64_t)t->dim, seed,
                                                                                     (double)sigma);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(ctx->stream));
    t->stats_valid = false;
    t->screen_overflow_streak = t->screen_backoff = 0;
    t->nx_valid = false;
    return PG_OK;
}

int pg_table_fill_mixture(pg_ctx* ctx, pg_table* t, uint64_t seed, uint32_t n_centres, float sigma) {
    PG_REQUIRE(ctx && t, "pg_table_fill_mixture: NULL argument");
    PG_REQUIRE(n_centres >= 1 && sigma >= 0.0f && sigma < 1e30f, "pg_table_fill_mixture: n_centres >= 1, sigma >= 0 and finite");
    PG_REQUIRE(t->dim == 64 || t->dim == 128, "pg_table_fill_mixture: dim=%u unsupported (64 or 128)", t->dim);
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableWrite w(t);
    const uint32_t blocks = (uint32_t)((t->rows + 255) / 256);
    const float ns = sigma * sqrtf(3.0f / (float)t->dim);
    if (t->dim == 64) pg::table_fill_mixture_kernel<64><<<blocks, 256, 0, ctx->stream>>>(t->d, t->rows, t->row_offset, seed, n_centres, ns);
    else pg::table_fill_mixture_kernel<128><<<blocks, 256, 0, ctx->stream>>>(t->d, t->rows, t->row_offset, seed, n_centres, ns);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(ctx->stream));
    t->stats_valid = false;
    t->screen_overflow_streak = t->screen_backoff = 0;
    t->nx_valid = false;
    return PG_OK;
}

int pg_table_upload(pg_ctx* ctx, pg_table* t, uint64_t row0, uint64_t nrows, const float* host_rows) {
    PG_REQUIRE(ctx && t && (nrows == 0 || host_rows), "pg_table_upload: NULL argument");
    PG_REQUIRE(row0 + nrows <= t->rows, "pg_table_upload: rows [%llu,%llu) outside table of %llu",
               (unsigned long long)row0, (unsigned long long)(row0 + nrows), (unsigned long long)t->rows);
    std::lock_guard<std::mutex> g(ctx->mu);
    if (nrows == 0) return PG_OK;
    pg::TableWrite w(t);                             // no enqueue reads the table's pointers meanwhile (rows of a table that is
                                                     // being served change under batches already in flight: load into a
                                                     // second table and pg_table_swap for an atomic change-over)
    PG_HIP(hipMemcpyAsync(t->d + row0 * t->dim, host_rows, nrows * (size_t)t->dim * sizeof(float),
                          hipMemcpyHostToDevice, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    t->stats_valid = false;
    t->screen_overflow_streak = t->screen_backoff = 0;
    t->nx_valid = false;
    return PG_OK;
}

int pg_table_download(pg_ctx* ctx, const pg_table* t, uint64_t row0, uint64_t nrows, float* host_rows) {
    PG_REQUIRE(ctx && t && (nrows == 0 || host_rows), "pg_table_download: NULL argument");
    PG_REQUIRE(row0 + nrows <= t->rows, "pg_table_download: rows out of range");
    std::lock_guard<std::mutex> g(ctx->mu);
    if (nrows == 0) return PG_OK;
    PG_HIP(hipMemcpyAsync(host_rows, t->d + row0 * t->dim, nrows * (size_t)t->dim * sizeof(float),
                          hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_table_swap(pg_ctx* ctx, pg_table* a, pg_table* b) {
    PG_REQUIRE(ctx && a && b, "pg_table_swap: NULL argument");
    PG_REQUIRE(a->rows == b->rows && a->dim == b->dim, "pg_table_swap: shapes differ");
    std::lock_guard<std::mutex> g(ctx->mu);          // no call on this context is mid-flight
    pg::TableWrite w(a, b);                          // ... and no other context is enqueueing against either table
    // Everything already enqueued holds the old pointers and stays valid (both buffers live on), but whoever swapped
    // will soon refill the table that now holds the old rows: drain the device so that nothing still reads them.
    PG_HIP(hipSetDevice(ctx->device));
    PG_HIP(hipDeviceSynchronize());
    std::swap(a->d, b->d);
    std::swap(a->row_offset, b->row_offset);
    std::swap(a->d_row_map, b->d_row_map);
    std::swap(a->map_offset, b->map_offset);
    std::swap(a->stats_valid, b->stats_valid);
    std::swap(a->all_finite, b->all_finite);
    std::swap(a->max_norm, b->max_norm);
    std::swap(a->d16, b->d16);
    std::swap(a->d8, b->d8);
    std::swap(a->dnorm2, b->dnorm2);
    std::swap(a->shadow_is_i8, b->shadow_is_i8);
    std::swap(a->s8, b->s8);
    std::swap(a->resid8, b->resid8);
    std::swap(a->shadow_failed, b->shadow_failed);
    std::swap(a->d4, b->d4);
    std::swap(a->d4s, b->d4s);
    std::swap(a->i4_ok, b->i4_ok);
    std::swap(a->i4_failed, b->i4_failed);
    std::swap(a->rho4, b->rho4);
    std::swap(a->rmax4, b->rmax4);
    std::swap(a->lam4, b->lam4);
    std::swap(a->i4m_pairs, b->i4m_pairs);
    std::swap(a->rec_scale, b->rec_scale);
    std::swap(a->d8r, b->d8r);
    std::swap(a->s8r, b->s8r);
    std::swap(a->resid2, b->resid2);
    std::swap(a->r2_ok, b->r2_ok);
    std::swap(a->r2_failed, b->r2_failed);
    std::swap(a->wide_susp, b->wide_susp);
    std::swap(a->prefix_failures, b->prefix_failures);
    std::swap(a->d_pred, b->d_pred);
    std::swap(a->d_nx, b->d_nx);
    std::swap(a->d_nxmin, b->d_nxmin);
    std::swap(a->l2_slack, b->l2_slack);
    std::swap(a->nx_valid, b->nx_valid);
    std::swap(a->pred_model, b->pred_model);
    std::swap(a->pred_k, b->pred_k);
    std::swap(a->pred_n, b->pred_n);
    std::swap(a->pred_sum, b->pred_sum);
    std::swap(a->pred_sum2, b->pred_sum2);
    std::swap(a->pred_min, b->pred_min);
    std::swap(a->pred_total, b->pred_total);
    std::swap(a->pred_backoff, b->pred_backoff);
    std::swap(a->pred_failures, b->pred_failures);
    std::swap(a->screen_overflow_streak, b->screen_overflow_streak);
    std::swap(a->screen_backoff, b->screen_backoff);
    return PG_OK;
}

int pg_table_info(const pg_table* t, uint64_t* rows, uint32_t* dim, uint64_t* row_offset) {
    PG_REQUIRE(t, "pg_table_info: table is NULL");
    if (rows) *rows = t->rows;
    if (dim) *dim = t->dim;
    if (row_offset) *row_offset = t->row_offset;
    return PG_OK;
}

int pg_table_gather(pg_ctx* ctx, const pg_table* t, const uint32_t* rows, uint32_t n, float* out) {
    PG_REQUIRE(ctx && t && (n == 0 || (rows && out)), "pg_table_gather: NULL argument");
    if (n == 0) return PG_OK;
    for (uint32_t i = 0; i < n; ++i)
        PG_REQUIRE(rows[i] < t->rows, "pg_table_gather: row %u out of range", rows[i]);
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::TableRead tr(t->rw);
    void *d_rows, *d_out;
    int rc;
    if ((rc = pg::scratch_reserve(ctx, 0, (size_t)n * 4, &d_rows))) return rc;
    if ((rc = pg::scratch_reserve(ctx, 1, (size_t)n * t->dim * 4, &d_out))) return rc;
    PG_HIP(hipMemcpyAsync(d_rows, rows, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t threads = (uint64_t)n * (t->dim / 4);
    pg::table_gather_kernel<<<(uint32_t)((threads + 255) / 256), 256, 0, ctx->stream>>>(
        t->d, t->dim, (const uint32_t*)d_rows, n, (float*)d_out);
    PG_HIP(hipGetLastError());
    PG_HIP(hipMemcpyAsync(out, d_out, (size_t)n * t->dim * 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

}  // extern "C"
