// Microtest: which halves v_permlane32_swap_b32 exchanges (gfx950), and the two one-way transfers built on it.
// Build: hipcc -O3 --offload-arch=gfx950 -o permlane_swap permlane_swap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
    unsigned c = 200 + threadIdx.x;
    asm volatile("" : "+v"(c));
    auto lo = __builtin_amdgcn_permlane32_swap(c, 0u, false, false);      // r[1]: lower lanes <- c's upper half?
    o[128 + threadIdx.x] = lo[1];
    unsigned d = 300 + threadIdx.x;
    asm volatile("" : "+v"(d));
    auto up = __builtin_amdgcn_permlane32_swap(0u, d, false, false);      // r[0]: upper lanes <- d's lower half?
    o[192 + threadIdx.x] = up[0];
    // the same inside a dependent chain, as the kernel uses them
    float x = (float)threadIdx.x;
    asm volatile("" : "+v"(x));
    for (int g = 0; g < 3; ++g) {
        auto t = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), 0u, false, false);
        x = __builtin_bit_cast(float, t[1]) + 1000.0f;
        auto u = __builtin_amdgcn_permlane32_swap(0u, __builtin_bit_cast(unsigned, x), false, false);
        x = __builtin_bit_cast(float, u[0]) + 10.0f;
    }
    o[256 + threadIdx.x] = (unsigned)x;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 2048);
    k<<<1, 64>>>(d);
    unsigned h[320];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l : {0, 31, 32, 63})
        printf("lane %2d: swap(a,b) r0 = %3u r1 = %3u | swap(c,0).r1 = %3u | swap(0,d).r0 = %3u | chain = %u\n", l, h[l], h[64 + l], h[128 + l], h[192 + l], h[256 + l]);
    return 0;
}
