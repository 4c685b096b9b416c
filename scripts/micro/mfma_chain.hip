// Issue interval of DEPENDENT bf16 MFMAs (the same accumulator as C and D) against independent ones, per wave:
// v_mfma_f32_32x32x16_bf16 (8 passes) with 1 / 2 / 4 accumulators in rotation, v_mfma_f32_16x16x32_bf16 (4 passes) with 1 / 4,
// and the same with 1, 2 or 3 waves per SIMD.  Cycles per MFMA from s_memtime around 256 MFMAs.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void chain32(uint64_t* out, float* sink) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = (float)threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)0.5f; }
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 256 / (4 * NACC); ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1.2345f) sink[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int NACC>
__global__ void chain16(uint64_t* out, float* sink) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (float)threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)0.5f; }
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 256 / (4 * NACC); ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1.2345f) sink[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <typename K>
static void run(K k, int waves, const char* tag, uint64_t* d_out, float* sink) {
    uint64_t h[16];
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 0, 0, d_out, sink);
    hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 0, 0, d_out, sink);
    hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    double m = 0;
    for (int w = 0; w < waves; ++w) m += (double)h[w] / waves;
    printf("%-44s %2d waves/CU: %6.1f cycles per MFMA per wave (%.1f per SIMD slot)\n", tag, waves, m / 256.0, m / 256.0 / ((waves + 3) / 4));
}
int main() {
    uint64_t* d_out; float* sink;
    hipMalloc(&d_out, 16 * 8); hipMalloc(&sink, 4);
    for (int waves : {4, 8, 12}) {
        run(chain32<1>, waves, "32x32x16, one accumulator (dependent chain)", d_out, sink);
        run(chain32<2>, waves, "32x32x16, two accumulators", d_out, sink);
        run(chain32<4>, waves, "32x32x16, four accumulators", d_out, sink);
        run(chain16<1>, waves, "16x16x32, one accumulator (dependent chain)", d_out, sink);
        run(chain16<4>, waves, "16x16x32, four accumulators", d_out, sink);
    }
    return 0;
}
