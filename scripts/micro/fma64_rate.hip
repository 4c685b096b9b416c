// Microbenchmark: v_fmac_f64 rate per SIMD with 1 / 2 / 4 waves per SIMD (64 independent accumulators per lane, the DPP
// kernel matrix's operand pattern: 4 + 4 operand pairs per 64 fma).  Prints T fma/s and cycles per wave-instruction.
// Build: hipcc -O3 --offload-arch=gfx950 -o fma64_rate fma64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WPS>
__global__ __launch_bounds__(64 * 4 * WPS) void k(double* __restrict__ out, int iters, double seed) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    double acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.0;
    double av[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        av[i] = seed + threadIdx.x * 1e-3 + i;
        bv[i] = seed - threadIdx.x * 1e-3 - i;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(av[i]), "+v"(bv[i]));
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
    }
    double s = 0;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) s += acc[a][b];
    if (s == 1.2345) out[threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        out[1000] = (double)(__builtin_readcyclecounter() - c0);
        out[1001] = (double)(wall_clock64() - r0);
    }
}

template <int WPS>
void run(double* d, int cus) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<WPS><<<cus, 64 * 4 * WPS>>>(d, 100, 1.0);
    hipEventRecord(e0);
    k<WPS><<<cus, 64 * 4 * WPS>>>(d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)cus * 4 * WPS * 64 * 64.0 * iters;
    double h[2];
    hipMemcpy(h, d + 1000, 16, hipMemcpyDeviceToHost);
    const double ghz = h[0] / (h[1] / 0.1);                 // wall_clock64 ticks at 100 MHz
    printf("%d wave(s) per SIMD: %.3f ms, %.2f T fma/s (%.1f TFLOP/s); shader clock %.2f GHz -> %.2f cycles per wave-instruction per SIMD\n", WPS, ms,
           fma / ms / 1e9, 2 * fma / ms / 1e9, ghz, (ms * 1e-3 * ghz * 1e9) / ((double)iters * 64 * WPS));
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    double* d;
    hipMalloc(&d, 1 << 20);
    run<1>(d, p.multiProcessorCount);
    run<2>(d, p.multiProcessorCount);
    return 0;
}
