// Microbenchmark: how fast does one wave per SIMD issue v_mfma_f32_32x32x16_bf16 with the operand pattern of
// the 256-query screen (8 accumulators in VGPRs, B operand parked in AGPRs, A operand from VGPRs)?
// Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NQB, bool B_IN_AGPR, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const uint4* __restrict__ bsrc, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    uint4 b[NQB][8];
#pragma unroll
    for (int c = 0; c < NQB; ++c)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            b[c][ks] = bsrc[(c * 8 + ks) * 64 + lane];
            if (B_IN_AGPR) asm volatile("" : "+a"(b[c][ks].x), "+a"(b[c][ks].y), "+a"(b[c][ks].z), "+a"(b[c][ks].w));
            else asm volatile("" : "+v"(b[c][ks].x), "+v"(b[c][ks].y), "+v"(b[c][ks].z), "+v"(b[c][ks].w));
        }
    f32x4 a[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a[ks] = f32x4{1.0f * lane, 2.0f, 3.0f, (float)ks};
    float sink = 0.0f;
    for (int it = 0; it < iters; ++it) {
        f32x16 acc[NQB];
#pragma unroll
        for (int c = 0; c < NQB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            asm volatile("" : "+v"(a[ks]));          // keep the A fragments opaque
#pragma unroll
            for (int c = 0; c < NQB; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ks]),
                                                                 __builtin_bit_cast(bf16x8, b[c][ks]), acc[c], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < NQB; ++c) sink += acc[c][0] + acc[c][15];
    }
    if (sink == 12345.678f) out[threadIdx.x] = sink;
}

template <int NQB, bool AG, int WAVES>
static void run(const char* name, const uint4* b, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NQB, AG, WAVES><<<256, 64 * WAVES>>>(b, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NQB, AG, WAVES><<<256, 64 * WAVES>>>(b, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * WAVES * iters * NQB * 8;
    const double flops = mfma * 32768.0;
    // cycles per MFMA per SIMD assuming WAVES/4 waves per SIMD
    printf("%-34s %8.3f ms  %7.1f TFLOP/s  %5.1f ns per MFMA per SIMD\n", name, ms, flops / ms / 1e9,
           ms * 1e6 / (iters * NQB * 8.0 * (WAVES / 4.0)));
}

int main() {
    uint4* b;
    float* out;
    hipMalloc(&b, 8 * 8 * 64 * sizeof(uint4));
    hipMemset(b, 0x3c, 8 * 8 * 64 * sizeof(uint4));
    hipMalloc(&out, 4096);
    run<8, true, 4>("NQB=8 B in AGPR, 1 wave/SIMD", b, out);
    run<4, false, 4>("NQB=4 B in VGPR, 1 wave/SIMD", b, out);
    run<4, false, 8>("NQB=4 B in VGPR, 2 waves/SIMD", b, out);
    run<2, false, 8>("NQB=2 B in VGPR, 2 waves/SIMD", b, out);
    run<8, true, 4>("NQB=8 B in AGPR, 1 wave/SIMD (again)", b, out);
    return 0;
}
