// rank_mlp.hpp — what the rank-stage kernels share: operand types, the bf16 rounding used on both sides of
// the boundary, the kernel argument block and the swizzled LDS operand-tile stores (rank_mlp.hip, rank_ws.hip).
#pragma once
#include "common.hpp"

#include <cstring>

namespace pg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
    uint32_t b;
#ifdef __HIP_DEVICE_COMPILE__
    b = __float_as_uint(x);
#else
    memcpy(&b, &x, 4);
#endif
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((b >> 16) | 0x0040u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}
__host__ __device__ __forceinline__ float bf16_to_f32(uint16_t v) {
    uint32_t b = (uint32_t)v << 16;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}
// operand rounding of a precision mode: 0 = fp32, 1 = bf16 (RNE), 2 = split bf16 (x = hi + lo, three MFMA products:
// nothing is rounded on the scalar side — the mode's specification is the fp32 one)
__host__ __device__ __forceinline__ float round_prec(float x, int prec) {
    return prec == 1 ? bf16_to_f32(f32_to_bf16_rne(x)) : x;
}

// PG_PREC_BF16X3: a pair of fp32 values as two packed bf16 pairs, hi = RNE(x), lo = RNE(x - hi) — x - hi is exact in
// fp32, so |x - hi - lo| <= 2^-17 |x|; with the weights split the same way, hi*hi + hi*lo + lo*hi leaves a relative
// error of ~2^-16 per product (the dropped lo*lo term), at fp32 accumulation.  A non-finite or > bf16-max value makes
// hi infinite and lo NaN: such inputs score NaN in this mode.
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_));            // v_cvt_pk_bf16_f32 (RNE)
    const f32x2_ r = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_));
}

constexpr int kBM = 128;       // items per workgroup tile
constexpr int kDIN = 128;      // gathered input width
constexpr int kFmMaxK = 32;        // largest FM embedding width (item fields x width = kDIN)
constexpr int kFmUserStride = 1 + 2 * kFmMaxK;   // per-request FM prefix: lin, s[<=32] at +1, q[<=32] at +33
constexpr int kMaxHeads = 8;           // outputs of a multi-head DNN3
constexpr int kItemRowFloats = 160;    // a materialised item record: 128 embedding floats + <= 16 linear weights, padded to 5 x 128 B

// the item-field columns of a feature store, by value (kernel argument of the item-record builder)
struct ItemRowCols {
    const void* base[16];
    double def[16];
    int32_t dtype[16];
};

struct MlpArgs {
    const uint32_t* tile_req;
    const uint32_t* tile_item0;
    const uint32_t* tile_cnt;
    const uint32_t* n_tiles;
    // DNN3 gather
    const float* tab;
    uint32_t tab_rows;
    uint32_t tab_dim;                // 64 or 128 floats per row; columns beyond it read as 0 (their W1 rows are 0 too)
    const uint32_t* cand_rows;
    // two-tower gather
    const float* const* field_emb;   // device array [n_user_fields + n_item_fields] of [vocab][k]
    const float* const* field_lin;   // device array [...] of [vocab]
    const int32_t* item_field_ids;   // [n_items][n_item_fields]
    uint32_t vocab;
    uint32_t n_user_fields;          // the item fields' tables follow the user fields' in field_emb / field_lin
    const float* fm_user;            // [n_req][kFmUserStride]: linU, sU[k] at +1, qU[k] at +33
    // two-tower gather from materialised item records (MODEL 3): record r = [kDIN embedding floats | n_item_fields
    // linear weights | pad] at irows + r * kItemRowFloats; candidates at or past irow_count read record irow_count
    // (the columns' defaults)
    const float* irows;
    uint32_t irow_count;
    // per request / shared vectors
    const float* c1;
    uint32_t c1_stride;
    const float* w3;
    uint32_t w3_stride;
    float b3;
    const float* b2;
    // DNN3 with several heads on the shared trunk: w3 is [n_out][h2], b3v [n_out] (b3 = b3v[0]); head o's scores go to
    // out + o * out_stride; head_part: global scratch for the weights-stationary kernel's partials of heads 1..
    uint32_t n_out;
    size_t out_stride;
    const float* b3v;
    float* head_part;
    // pre-packed weights (PG_PREC_BF16X3: the hi fragments; the lo fragments follow at w1p_lo / w2p_lo)
    const void* w1p;
    const void* w2p;
    const void* w1p_lo;
    const void* w2p_lo;
    float* out;
    float* sink;                     // >= 1024 floats nobody reads (fm2t_irs_kernel's always-issued stores)
};

// (PREC 2: the hi tile has the bf16 layout, the lo tile follows it `lo_off` bytes on)
template <int PREC>
__device__ __forceinline__ void store_x_quad(char* tile, int row, int c, float4 v, int lo_off = 0) {
    if constexpr (PREC == 2) {
        uint2 ph, pl;
        split_bf16x2(v.x, v.y, ph.x, pl.x);
        split_bf16x2(v.z, v.w, ph.y, pl.y);
        char* const d = tile + row * 256 + ((((c >> 1) ^ (row & 15))) << 4) + (c & 1) * 8;
        *reinterpret_cast<uint2*>(d) = ph;
        *reinterpret_cast<uint2*>(d + lo_off) = pl;
    } else if constexpr (PREC == 1) {
        // 4 bf16 = 8 B at element 4c: 16-B quad index c/2, XOR-swizzled by row
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
        uint2 p;
        p.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
        p.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2));
        *reinterpret_cast<uint2*>(tile + row * 256 + ((((c >> 1) ^ (row & 15))) << 4) + (c & 1) * 8) = p;
    } else {
        *reinterpret_cast<float4*>(tile + row * 512 + ((c ^ (row & 15)) << 4)) = v;
    }
}

// element (row, col) of an LDS operand tile with K columns per row: 16-B quads XOR-swizzled by row
template <int PREC, int K>
__device__ __forceinline__ void store_h_elem(char* tile, int row, int col, float v) {
    constexpr int ES = PREC ? 2 : 4;
    constexpr int ROWB = K * ES;
    constexpr int SW = (ROWB / 16 < 16 ? ROWB / 16 : 16) - 1;
    if constexpr (PREC == 1) {
        *reinterpret_cast<uint16_t*>(tile + row * ROWB + ((((col >> 3) ^ (row & SW))) << 4) + (col & 7) * 2) =
            f32_to_bf16_rne(v);
    } else {
        *reinterpret_cast<float*>(tile + row * ROWB + ((((col >> 2) ^ (row & SW))) << 4) + (col & 3) * 4) = v;
    }
}

// 4 consecutive columns col..col+3 (col % 4 == 0) of one row, after relu and operand rounding
template <int PREC, int K>
__device__ __forceinline__ void store_h_quad(char* tile, int row, int col, float v0, float v1, float v2, float v3, int lo_off = 0) {
    constexpr int ES = PREC ? 2 : 4;
    constexpr int ROWB = K * ES;
    constexpr int SW = (ROWB / 16 < 16 ? ROWB / 16 : 16) - 1;
    if constexpr (PREC == 2) {
        uint2 ph, pl;
        split_bf16x2(v0, v1, ph.x, pl.x);
        split_bf16x2(v2, v3, ph.y, pl.y);
        char* const d = tile + row * ROWB + ((((col >> 3) ^ (row & SW))) << 4) + (col & 7) * 2;
        *reinterpret_cast<uint2*>(d) = ph;
        *reinterpret_cast<uint2*>(d + lo_off) = pl;
    } else if constexpr (PREC == 1) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const f32x2 lo = {v0, v1}, hi = {v2, v3};
        uint2 p;
        p.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf16x2));   // v_cvt_pk_bf16_f32 (RNE)
        p.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf16x2));
        *reinterpret_cast<uint2*>(tile + row * ROWB + ((((col >> 3) ^ (row & SW))) << 4) + (col & 7) * 2) = p;
    } else {
        *reinterpret_cast<float4*>(tile + row * ROWB + ((((col >> 2) ^ (row & SW))) << 4)) = make_float4(v0, v1, v2, v3);
    }
}

// rank_ws.hip: the weights-stationary DNN3 kernel (bf16); `a` describes 64-item tiles
constexpr int kWsItems = 64;
int launch_dnn3_ws(pg_ctx* ctx, const MlpArgs& a);
// rank_rs.hip: the register-stationary DNN3 kernel (bf16) for the small hidden shapes; 64-item tiles as well
bool dnn3_rs_shape(uint32_t h1, uint32_t h2);
int launch_dnn3_rs(pg_ctx* ctx, uint32_t h1, uint32_t h2, const MlpArgs& a);
// ... and the eight-wave streamed-weights kernel for 1024-512; 128-item tiles
bool dnn3_ls_shape(uint32_t h1, uint32_t h2);
int launch_dnn3_ls(pg_ctx* ctx, uint32_t h1, uint32_t h2, const MlpArgs& a);

// rank_x3.hip: DNN3 in PG_PREC_BF16X3 (split bf16) — 128-item tiles, layer-1 and layer-2 waves sharing each SIMD
bool dnn3_x3_shape(uint32_t h1, uint32_t h2);
int launch_dnn3_x3(pg_ctx* ctx, uint32_t h1, uint32_t h2, const MlpArgs& a);

// rank_ir.hip: FM + two-tower over materialised item records for the benchmark's shape (towers 256-64, 8 fields x 16, bf16):
// weights stationary in registers, records two tiles ahead; 64-item tiles
constexpr int kIrsItems = 64;
bool fm2t_irs_shape(uint32_t th, uint32_t to, uint32_t k, uint32_t nif, int prec);
int launch_fm2t_irs(pg_ctx* ctx, const MlpArgs& a);
// the same shape with every wave a whole pipeline over 32-item tiles (rank_is.hip)
constexpr int kIswItems = 32;
int launch_fm2t_isw(pg_ctx* ctx, const MlpArgs& a);

}  // namespace pg
