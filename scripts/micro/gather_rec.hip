// Random gather of 640-B item records (the FM + two-tower model's materialised item side): records per second by
// footprint, lanes per record and occupancy.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gather_rec.hip -o /tmp/gather_rec && /tmp/gather_rec
// LPR lanes share a record: lane j reads the 16-B quads j, j + LPR, ... of its 34 useful quads (544 B).  LDSB bytes of
// dynamic LDS per workgroup limit the workgroups per CU the way the rank kernel's tiles do (66 KB -> two per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int LPR>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ buf, const uint32_t* __restrict__ rows, uint32_t n, float* out) {
    extern __shared__ char lds[];
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t rec = tid / LPR, j = tid % LPR;
    float acc = 0.f;
    if (rec < n) {
        const float4* p = buf + (size_t)rows[rec] * 40;
        constexpr int NQ = (34 + LPR - 1) / LPR;
        float4 v[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = j + i * LPR;
            v[i] = q < 34 ? p[q] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (acc == 1.2345f) { out[0] = acc; lds[0] = 1; }
}

template <int LPR>
static void run(const float4* buf, const uint32_t* rows, uint32_t n, float* out, size_t ldsb, const char* tag) {
    const uint32_t blocks = (uint32_t)(((uint64_t)n * LPR + 255) / 256);
    hipFuncSetAttribute((const void*)gather<LPR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    gather<LPR><<<blocks, 256, ldsb>>>(buf, rows, n, out);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        gather<LPR><<<blocks, 256, ldsb>>>(buf, rows, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("  %-26s lanes/record %2d  %7.3f ms  %6.2f G records/s  %5.2f TB/s of 128-B lines\n", tag, LPR, best, n / best / 1e6, n * 640.0 / best / 1e9);
}

int main() {
    const uint32_t n = 1280000;
    float* out;
    hipMalloc(&out, 4);
    for (uint64_t recs : {1000000ull, 4000000ull, 20000000ull, 80000000ull}) {
        const uint64_t bytes = recs * 640;
        float4* buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("cannot allocate %llu bytes\n", (unsigned long long)bytes); continue; }
        hipMemset(buf, 0, bytes);
        uint32_t* h = (uint32_t*)malloc((size_t)n * 4);
        uint64_t x = 88172645463325252ull;
        for (uint32_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x % recs); }
        uint32_t* rows;
        hipMalloc(&rows, (size_t)n * 4);
        hipMemcpy(rows, h, (size_t)n * 4, hipMemcpyHostToDevice);
        printf("catalogue of %llu records (%.1f GB), %u random candidates\n", (unsigned long long)recs, bytes / 1e9, n);
        for (size_t ldsb : {(size_t)0, (size_t)66 * 1024}) {
            const char* tag = ldsb ? "2 workgroups per CU" : "full occupancy";
            run<1>(buf, rows, n, out, ldsb, tag);
            run<2>(buf, rows, n, out, ldsb, tag);
            run<8>(buf, rows, n, out, ldsb, tag);
            run<16>(buf, rows, n, out, ldsb, tag);
        }
        hipFree(rows); hipFree(buf); free(h);
    }
    return 0;
}
