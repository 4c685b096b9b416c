// Random 128-B row gather rate (the int8 stage of recall_i4m.hip: one int8 shadow row per suspect).  Eight lanes read a row
// (16 B each), NF rows per lane group in flight; rows at pseudo-random 128-B slots of a 12.8 GB buffer.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gather128.hip -o scripts/micro/gather128 && scripts/micro/gather128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NF, int LPR>      // LPR lanes per row (8: 128 B, 4 with 32 B each is not expressible as one load: 8 or 16 lanes x 16 / 8 B)
__global__ __launch_bounds__(256) void gather(const u32x4* __restrict__ buf, uint64_t slots, uint32_t rounds, uint32_t* out) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t x = (tid / LPR) * 0x9E3779B97F4A7C15ull + 12345;
    const int j = threadIdx.x % LPR;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < rounds; ++i) {
        u32x4 v[NF];
#pragma unroll
        for (int r = 0; r < NF; ++r) {
            x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
            v[r] = __builtin_nontemporal_load(buf + (x % slots) * 8 + j);
        }
#pragma unroll
        for (int r = 0; r < NF; ++r) acc += v[r].x ^ v[r].y ^ v[r].z ^ v[r].w;
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <int NF>
static void run(const u32x4* buf, uint64_t bytes, uint32_t* out, uint32_t blocks) {
    const uint64_t slots = bytes / 128;
    const uint32_t rounds = 256 / NF;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        gather<NF, 8><<<blocks, 256>>>(buf, slots, rounds, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    const double rows = (double)blocks * 256 / 8 * rounds * NF;
    printf("128-B rows, 8 lanes per row, %2d in flight per lane, %5u blocks: %8.3f ms  %6.2f G rows/s  %5.2f TB/s\n", NF, blocks, best,
           rows / best / 1e6, rows * 128 / best / 1e9);
}

int main() {
    const uint64_t bytes = 12800ull << 20;
    u32x4* buf; uint32_t* out;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    for (uint32_t blocks : {256u * 4, 256u * 8, 256u * 16, 256u * 32}) {
        run<4>(buf, bytes, out, blocks);
        run<8>(buf, bytes, out, blocks);
        run<16>(buf, bytes, out, blocks);
        run<32>(buf, bytes, out, blocks);
    }
    return 0;
}
