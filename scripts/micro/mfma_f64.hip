// v_mfma_f64_16x16x4_f64: (1) does D = C + sum_k A[i][k] B[k][j] round like the k-ascending fma chain the DPP kernel matrix is
// specified as (oracle/oracle.c orc_dpp_kernel_matrix_f: acc = fma(a_k, b_k, acc), k ascending)?  16 x 16 outputs over K = 128
// (32 chained instructions) on operands whose exponents spread over 2^-20 .. 2^20, compared bit for bit with four host
// candidates: the ascending chain, the descending chain, pairwise inside each instruction, and products summed in long double.
// (2) its issue interval: cycles per instruction with 1 / 2 / 4 accumulators in rotation and 1 / 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_f64.hip -o /tmp/mfma_f64 && /tmp/mfma_f64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int K = 128;
// A [16][K], B [K][16] row-major in global memory; one wave
__global__ void gemm16(const double* A, const double* B, double* D) {
    const int lane = threadIdx.x & 63;
    f64x4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = A[(lane & 15) * K + k0 + (lane >> 4)];       // A[i = lane & 15][k = lane >> 4]
        const double b = B[(k0 + (lane >> 4)) * 16 + (lane & 15)];    // B[k = lane >> 4][j = lane & 15]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];   // row = (lane >> 4) + 4 r, col = lane & 15
}

template <int NACC>
__global__ void rate(uint64_t* out, double* sink) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (double)threadIdx.x;
    double a = 1.0 + threadIdx.x, b = 0.5;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 256 / (4 * NACC); ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const uint64_t t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1.2345) sink[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <typename Kn>
static void run(Kn k, int waves, const char* tag, uint64_t* d_out, double* sink) {
    uint64_t h[16];
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 0, 0, d_out, sink);
    hipDeviceSynchronize();
    hipMemcpy(h, d_out, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost);
    printf("%-28s %d wave(s)/SIMD: s_memtime ticks per instruction, wave by wave:", tag, (waves + 3) / 4);
    for (int w = 0; w < waves; ++w) printf(" %.1f", (double)h[w] / 256.0);
    printf("\n");
}
// wall-clock rate: every SIMD of the chip runs `waves` waves of 4096 instructions each
template <int NACC>
__global__ void rate_long(double* sink) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (double)threadIdx.x;
    double a = 1.0 + threadIdx.x, b = 0.5;
#pragma unroll 1
    for (int it = 0; it < 4096 / (4 * NACC); ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 1.2345) sink[0] = s;
}
static void wall(int waves_per_simd, double* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;                                  // 8 workgroups per CU's worth of work
    hipLaunchKernelGGL(rate_long<4>, dim3(blocks), dim3(waves_per_simd * 256), 0, 0, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_long<4>, dim3(blocks), dim3(waves_per_simd * 256), 0, 0, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * waves_per_simd * 4 * 4096;
    printf("wall clock, %d wave(s) per SIMD resident: %.3f ms for %.3g instructions = %.1f T fma/s = %.1f ns per instruction and SIMD\n",
           waves_per_simd, ms, insts, insts * 1024 / (ms * 1e-3) / 1e12, ms * 1e6 / (insts / 1024.0));
}

int main() {
    std::vector<double> A(16 * K), B(K * 16), D(256);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&] {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double m = 1.0 + (double)(s >> 12) / 4503599627370496.0;                 // [1, 2): 52 random mantissa bits
        const int e = (int)((s >> 3) % 41) - 20;
        return ((s & 1) ? -m : m) * std::ldexp(1.0, e);
    };
    for (auto& x : A) x = rnd();
    for (auto& x : B) x = rnd();
    double *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dD, 256 * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gemm16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    int eq_asc = 0, eq_desc = 0, eq_pair = 0, eq_exact = 0;
    double worst = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double asc = 0, pair = 0;
            long double ex = 0;
            for (int k = 0; k < K; ++k) asc = std::fma(A[i * K + k], B[k * 16 + j], asc);
            double desc = 0;
            for (int k0 = 0; k0 < K; k0 += 4) {                                         // instruction order kept, inside it descending
                for (int k = k0 + 3; k >= k0; --k) desc = std::fma(A[i * K + k], B[k * 16 + j], desc);
                const double p01 = std::fma(A[i * K + k0], B[k0 * 16 + j], A[i * K + k0 + 1] * B[(k0 + 1) * 16 + j]);
                const double p23 = std::fma(A[i * K + k0 + 2], B[(k0 + 2) * 16 + j], A[i * K + k0 + 3] * B[(k0 + 3) * 16 + j]);
                pair = pair + (p01 + p23);
            }
            for (int k = 0; k < K; ++k) ex += (long double)A[i * K + k] * (long double)B[k * 16 + j];
            const double got = D[i * 16 + j];
            eq_asc += memcmp(&got, &asc, 8) == 0;
            eq_desc += memcmp(&got, &desc, 8) == 0;
            eq_pair += memcmp(&got, &pair, 8) == 0;
            const double exd = (double)ex;
            eq_exact += memcmp(&got, &exd, 8) == 0;
            worst = std::fmax(worst, std::fabs(got - asc) / std::fmax(std::fabs(asc), 1e-300));
        }
    printf("v_mfma_f64_16x16x4_f64 over K = %d, 256 outputs: bit-equal to the k-ascending fma chain %d, to the per-instruction "
           "descending chain %d, to pairwise %d, to the long-double sum %d; worst relative difference to the ascending chain %.3g\n",
           K, eq_asc, eq_desc, eq_pair, eq_exact, worst);
    uint64_t* d_out;
    double* sink;
    hipMalloc(&d_out, 16 * 8); hipMalloc(&sink, 8);
    run(rate<1>, 4, "dependent (1 accumulator)", d_out, sink);
    run(rate<2>, 4, "2 accumulators", d_out, sink);
    run(rate<4>, 4, "4 accumulators", d_out, sink);
    run(rate<1>, 8, "dependent (1 accumulator)", d_out, sink);
    run(rate<4>, 8, "4 accumulators", d_out, sink);
    wall(1, sink);
    wall(2, sink);
    return 0;
}
