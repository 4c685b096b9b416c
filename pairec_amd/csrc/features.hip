// features.hip — device-resident typed columnar item features and their assembly into model inputs.
//
// In the reference every rank request re-boxes features on the host: EasyrecAlgoDataGenerator.AddFeatures
// (service/rank/algo_data.go:223-271) walks map[string]interface{} per item, keeps one []interface{} column
// per feature name ("context features", easyrec_predict.proto:150-212 PBFeature / ContextFeatures), and
// fills an item that lacks a feature with the Go zero value of the column's type (feature.defaultValue,
// algo_data.go:154-171).  Here the columns live in HBM once, keyed by item row, and a request only names
// candidate rows: assembly is a gather.  An item without the feature is a row index past the store
// (UINT32_MAX by convention) and reads the column default.
//
// Numeric columns only (int32 / int64 / float32 / float64, the PBFeature scalar kinds); string features are
// dictionary-encoded to integer ids by the host before upload.  The "simple normalizer" on the float path is
// value*scale + bias in fp32 (one fmaf); anything richer goes through the expression evaluator (pg_expr_*).
#include "common.hpp"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace pg {

struct ColDesc {
    const void* base;
    int32_t dtype;
    int32_t pad;
    double def;
};

__device__ __forceinline__ double load_as_f64(const ColDesc& c, uint32_t row, uint64_t rows) {
    if (row >= rows) return c.def;
    switch (c.dtype) {
        case PG_F_I32: return (double)((const int32_t*)c.base)[row];
        case PG_F_I64: return (double)((const int64_t*)c.base)[row];
        case PG_F_F32: return (double)((const float*)c.base)[row];
        default: return ((const double*)c.base)[row];
    }
}

// out[i][f] = int32 view of integer column f at rows[i] (int64 values saturate; default for absent rows)
__global__ void features_gather_i32_kernel(const ColDesc* __restrict__ cols, uint32_t F, uint64_t rows,
                                           const uint32_t* __restrict__ cand, uint32_t n,
                                           int32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * F) return;
    const uint32_t i = t / F, f = t % F;
    const ColDesc c = cols[f];
    const uint32_t row = cand[i];
    int64_t v;
    if (row >= rows) v = (int64_t)c.def;
    else v = c.dtype == PG_F_I32 ? (int64_t)((const int32_t*)c.base)[row] : ((const int64_t*)c.base)[row];
    v = v > 2147483647ll ? 2147483647ll : (v < -2147483648ll ? -2147483648ll : v);
    out[t] = (int32_t)v;
}

// out[i][f] = fmaf((float)value, scale[f], bias[f])   (scale = 1, bias = 0 when the arrays are NULL)
__global__ void features_gather_f32_kernel(const ColDesc* __restrict__ cols, uint32_t F, uint64_t rows,
                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                           const uint32_t* __restrict__ cand, uint32_t n,
                                           float* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * F) return;
    const uint32_t i = t / F, f = t % F;
    const float v = (float)load_as_f64(cols[f], cand[i], rows);
    out[t] = __fmaf_rn(v, scale ? scale[f] : 1.0f, bias ? bias[f] : 0.0f);
}

static size_t dtype_size(int dt) {
    switch (dt) {
        case PG_F_I32: case PG_F_F32: return 4;
        case PG_F_I64: case PG_F_F64: return 8;
    }
    return 0;
}

// descriptors of the requested columns (+ optional per-column scale / bias) → device scratch slot 1
static int stage_descs(pg_ctx* ctx, const pg_features* fs, const int32_t* col_idx, uint32_t F, bool ints_only,
                       const char* who, const float* scale, const float* bias, ColDesc** d_desc,
                       float** d_scale, float** d_bias) {
    std::vector<ColDesc> h(F);
    for (uint32_t f = 0; f < F; ++f) {
        if (col_idx[f] < 0 || (size_t)col_idx[f] >= fs->cols.size()) {
            set_error("%s: column index %d out of range (%zu columns)", who, col_idx[f], fs->cols.size());
            return PG_ERR_INVALID;
        }
        const auto& c = fs->cols[(size_t)col_idx[f]];
        if (ints_only && c.dtype != PG_F_I32 && c.dtype != PG_F_I64) {
            set_error("%s: column \"%s\" is not an integer column", who, c.name.c_str());
            return PG_ERR_INVALID;
        }
        h[f] = ColDesc{c.d, c.dtype, 0, c.def};
    }
    const size_t desc_bytes = ((size_t)F * sizeof(ColDesc) + 255) & ~(size_t)255;
    void* p;
    int rc;
    if ((rc = scratch_reserve(ctx, 1, desc_bytes + (size_t)F * 8 + 256, &p))) return rc;
    PG_HIP(hipMemcpyAsync(p, h.data(), (size_t)F * sizeof(ColDesc), hipMemcpyHostToDevice, ctx->stream));
    float* aux = (float*)((char*)p + desc_bytes);
    *d_scale = nullptr;
    *d_bias = nullptr;
    if (scale) {
        *d_scale = aux;
        PG_HIP(hipMemcpyAsync(aux, scale, (size_t)F * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if (bias) {
        *d_bias = aux + F;
        PG_HIP(hipMemcpyAsync(aux + F, bias, (size_t)F * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    PG_HIP(hipStreamSynchronize(ctx->stream));       // the host staging buffers go out of scope
    *d_desc = (ColDesc*)p;
    return PG_OK;
}

// up to 16 columns (an FM model's item fields): the descriptors travel as kernel arguments — nothing is staged, nothing
// synchronises, so the gather can sit inside a batch that is only enqueued (coalescer.hip)
struct ColDescs16 { ColDesc c[16]; };
__global__ void features_gather_i32_args_kernel(ColDescs16 cols, uint32_t F, uint64_t rows, const uint32_t* __restrict__ cand,
                                                uint32_t n, int32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * F) return;
    const uint32_t i = t / F, f = t % F;
    const ColDesc c = cols.c[f];
    const uint32_t row = cand[i];
    int64_t v;
    if (row >= rows) v = (int64_t)c.def;
    else v = c.dtype == PG_F_I32 ? (int64_t)((const int32_t*)c.base)[row] : ((const int64_t*)c.base)[row];
    v = v > 2147483647ll ? 2147483647ll : (v < -2147483648ll ? -2147483648ll : v);
    out[t] = (int32_t)v;
}

int features_gather_i32_locked(pg_ctx* ctx, const pg_features* fs, const int32_t* col_idx, uint32_t n_cols,
                               const uint32_t* d_rows, uint32_t n, int32_t* d_out, const char* who) {
    if (n_cols <= 16) {
        ColDescs16 h;
        memset(&h, 0, sizeof h);
        for (uint32_t f = 0; f < n_cols; ++f) {
            if (col_idx[f] < 0 || (size_t)col_idx[f] >= fs->cols.size()) {
                set_error("%s: column index %d out of range (%zu columns)", who, col_idx[f], fs->cols.size());
                return PG_ERR_INVALID;
            }
            const auto& c = fs->cols[(size_t)col_idx[f]];
            if (c.dtype != PG_F_I32 && c.dtype != PG_F_I64) {
                set_error("%s: column \"%s\" is not an integer column", who, c.name.c_str());
                return PG_ERR_INVALID;
            }
            h.c[f] = ColDesc{c.d, c.dtype, 0, c.def};
        }
        const uint32_t total = n * n_cols;
        features_gather_i32_args_kernel<<<(total + 255) / 256, 256, 0, ctx->stream>>>(h, n_cols, fs->rows, d_rows, n, d_out);
        PG_HIP(hipGetLastError());
        return PG_OK;
    }
    ColDesc* d_desc;
    float *d_scale, *d_bias;
    int rc;
    if ((rc = stage_descs(ctx, fs, col_idx, n_cols, true, who, nullptr, nullptr, &d_desc, &d_scale, &d_bias))) return rc;
    const uint32_t total = n * n_cols;
    features_gather_i32_kernel<<<(total + 255) / 256, 256, 0, ctx->stream>>>(d_desc, n_cols, fs->rows, d_rows, n, d_out);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // namespace pg

extern "C" {

int pg_features_create(pg_ctx* ctx, uint64_t rows, pg_features** out) {
    PG_REQUIRE(ctx && out, "pg_features_create: NULL argument");
    PG_REQUIRE(rows > 0 && rows < 0xFFFFFFFFull, "pg_features_create: rows must be in (0, 2^32-1)");
    *out = new pg_features();
    (*out)->rows = rows;
    return PG_OK;
}

int pg_features_destroy(pg_ctx* ctx, pg_features* fs) {
    PG_REQUIRE(ctx, "pg_features_destroy: NULL context");
    if (!fs) return PG_OK;
    std::lock_guard<std::mutex> g(ctx->mu);
    PG_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& c : fs->cols)
        if (c.d) PG_HIP(hipFree(c.d));
    delete fs;
    return PG_OK;
}

int pg_features_set_column(pg_ctx* ctx, pg_features* fs, const char* name, int dtype, const void* host_values,
                           double default_value) {
    PG_REQUIRE(ctx && fs && name && name[0], "pg_features_set_column: NULL argument");
    const size_t es = pg::dtype_size(dtype);
    PG_REQUIRE(es != 0, "pg_features_set_column: unknown dtype %d", dtype);
    std::lock_guard<std::mutex> g(ctx->mu);
    pg_features::Column* col = nullptr;
    for (auto& c : fs->cols)
        if (c.name == name) col = &c;
    if (col && col->dtype != dtype) {                    // type change: reallocate
        PG_HIP(hipStreamSynchronize(ctx->stream));
        PG_HIP(hipFree(col->d));
        col->d = nullptr;
    }
    const bool is_new = col == nullptr;
    if (is_new) {
        fs->cols.emplace_back();
        col = &fs->cols.back();
        col->name = name;
    }
    col->dtype = dtype;
    col->def = default_value;
    if (!col->d) {
        hipError_t e = hipMalloc(&col->d, fs->rows * es);
        if (e != hipSuccess) {
            col->d = nullptr;
            if (is_new) fs->cols.pop_back();
            pg::set_error("pg_features_set_column: hipMalloc(%zu) failed: %s", (size_t)(fs->rows * es), hipGetErrorString(e));
            return PG_ERR_NOMEM;
        }
    }
    if (host_values) {
        PG_HIP(hipMemcpyAsync(col->d, host_values, fs->rows * es, hipMemcpyHostToDevice, ctx->stream));
    } else {
        // every row holds the default (the column exists, no item has a value yet)
        std::vector<uint8_t> fill(fs->rows * es);
        for (uint64_t r = 0; r < fs->rows; ++r) {
            switch (dtype) {
                case PG_F_I32: ((int32_t*)fill.data())[r] = (int32_t)default_value; break;
                case PG_F_I64: ((int64_t*)fill.data())[r] = (int64_t)default_value; break;
                case PG_F_F32: ((float*)fill.data())[r] = (float)default_value; break;
                default: ((double*)fill.data())[r] = default_value;
            }
        }
        PG_HIP(hipMemcpyAsync(col->d, fill.data(), fs->rows * es, hipMemcpyHostToDevice, ctx->stream));
        PG_HIP(hipStreamSynchronize(ctx->stream));
        return PG_OK;
    }
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_features_column_index(const pg_features* fs, const char* name) {
    if (!fs || !name) return -1;
    for (size_t i = 0; i < fs->cols.size(); ++i)
        if (fs->cols[i].name == name) return (int)i;
    return -1;
}

int pg_features_num_columns(const pg_features* fs) { return fs ? (int)fs->cols.size() : 0; }

int pg_features_gather_i32_dev(pg_ctx* ctx, const pg_features* fs, const int32_t* col_idx, uint32_t n_cols,
                               const uint32_t* d_rows, uint32_t n, int32_t* d_out) {
    PG_REQUIRE(ctx && fs && (n_cols == 0 || col_idx), "pg_features_gather_i32_dev: NULL argument");
    if (n == 0 || n_cols == 0) return PG_OK;
    PG_REQUIRE(d_rows && d_out, "pg_features_gather_i32_dev: NULL argument");
    PG_REQUIRE((uint64_t)n * n_cols < 0xFFFFFFFFull, "pg_features_gather_i32_dev: n x n_cols too large");
    std::lock_guard<std::mutex> g(ctx->mu);
    return pg::features_gather_i32_locked(ctx, fs, col_idx, n_cols, d_rows, n, d_out, "pg_features_gather_i32_dev");
}

int pg_features_gather_f32_dev(pg_ctx* ctx, const pg_features* fs, const int32_t* col_idx, uint32_t n_cols,
                               const float* scale, const float* bias, const uint32_t* d_rows, uint32_t n,
                               float* d_out) {
    PG_REQUIRE(ctx && fs && (n_cols == 0 || col_idx), "pg_features_gather_f32_dev: NULL argument");
    if (n == 0 || n_cols == 0) return PG_OK;
    PG_REQUIRE(d_rows && d_out, "pg_features_gather_f32_dev: NULL argument");
    PG_REQUIRE((uint64_t)n * n_cols < 0xFFFFFFFFull, "pg_features_gather_f32_dev: n x n_cols too large");
    std::lock_guard<std::mutex> g(ctx->mu);
    pg::ColDesc* d_desc;
    float *d_scale, *d_bias;
    int rc;
    if ((rc = pg::stage_descs(ctx, fs, col_idx, n_cols, false, "pg_features_gather_f32_dev", scale, bias, &d_desc,
                              &d_scale, &d_bias)))
        return rc;
    const uint32_t total = n * n_cols;
    pg::features_gather_f32_kernel<<<(total + 255) / 256, 256, 0, ctx->stream>>>(d_desc, n_cols, fs->rows, d_scale, d_bias,
                                                                            d_rows, n, d_out);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

}  // extern "C"
