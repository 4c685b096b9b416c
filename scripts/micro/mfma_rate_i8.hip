// Microbenchmark: issue rate of v_mfma_i32_32x32x32_i8 against v_mfma_f32_32x32x16_bf16, one wave per SIMD,
// 8 independent accumulators, B operand in AGPRs (the 256-query screen's operand pattern).
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_rate_i8 mfma_rate_i8.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool I8>
__global__ __launch_bounds__(256, 1) void k(const i32x4* __restrict__ bsrc, int* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    i32x4 b[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            b[c][ks] = bsrc[(c * 4 + ks) * 64 + lane];
            asm volatile("" : "+a"(b[c][ks]));
        }
    i32x4 a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a[ks] = i32x4{lane, 2, 3, ks};
    int sink = 0;
    for (int it = 0; it < iters; ++it) {
        if (I8) {
            i32x16 acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                asm volatile("" : "+v"(a[ks]));
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks], b[c][ks], acc[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) sink += acc[c][0] + acc[c][15];
        } else {
            f32x16 acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                asm volatile("" : "+v"(a[ks]));
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks]),
                                                                     __builtin_bit_cast(bf16x8, b[c][ks]), acc[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) sink += (int)(acc[c][0] + acc[c][15]);
        }
    }
    if (sink == 123456789) out[threadIdx.x] = sink;
}

template <bool I8>
static void run(const char* name, const i32x4* b, int* out) {
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<I8><<<256, 256>>>(b, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<I8><<<256, 256>>>(b, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * 4 * iters * 32;
    const double ops = mfma * 32.0 * 32.0 * (I8 ? 32 : 16) * 2;
    printf("%-30s %8.3f ms  %7.1f Tops/s  %5.1f ns per MFMA per SIMD\n", name, ms, ops / ms / 1e9, ms * 1e6 / (iters * 32.0));
}

int main() {
    i32x4* b;
    int* out;
    hipMalloc(&b, 8 * 4 * 64 * sizeof(i32x4));
    hipMemset(b, 0x01, 8 * 4 * 64 * sizeof(i32x4));
    hipMalloc(&out, 4096);
    run<false>("bf16 32x32x16", b, out);
    run<true>("i8 32x32x32", b, out);
    run<false>("bf16 32x32x16 (again)", b, out);
    run<true>("i8 32x32x32 (again)", b, out);
    return 0;
}
