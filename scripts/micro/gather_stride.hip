// Random gather of item records at two record strides — 640 B (five whole 128-B lines) and 576 B (4.5 lines: half the
// records start in the middle of a line) — 544 useful bytes each, four lanes per record (as fm2t_isw_kernel gathers).
// Does the memory side move 128-B lines (then the shorter stride buys nothing) or 64-B sectors (then it saves 10 %)?
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gather_stride.hip -o /tmp/gather_stride && /tmp/gather_stride
// (time per 1.28 M candidates; run under rocprofv3 --pmc FETCH_SIZE for the bytes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SQ>      // record stride in 16-B quads
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ buf, const uint32_t* __restrict__ rows, uint32_t n, float* out) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t rec = tid >> 2, j = tid & 3;
    float acc = 0.f;
    if (rec < n) {
        const float4* p = buf + (size_t)rows[rec] * SQ;
        float4 v[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = j + i * 4;
            v[i] = q < 34 ? p[q] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <int SQ>
static void run(const float4* buf, const uint32_t* rows, uint32_t n, float* out, const char* tag) {
    const uint32_t blocks = (uint32_t)(((uint64_t)n * 4 + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    gather<SQ><<<blocks, 256>>>(buf, rows, n, out);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        gather<SQ><<<blocks, 256>>>(buf, rows, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-28s %.4f ms  %.2f G records/s  %.2f TB/s of the %d B stride\n", tag, best, n / best / 1e6, (double)n * SQ * 16 / best / 1e9, SQ * 16);
}

int main() {
    const uint64_t n_rec = 20000000;
    const uint32_t n = 1280000;
    float4* buf; uint32_t* rows; float* out;
    hipMalloc(&buf, n_rec * 640); hipMalloc(&rows, n * 4); hipMalloc(&out, 4);
    hipMemset(buf, 0, n_rec * 640);
    std::vector<uint32_t> h(n);
    uint64_t x = 88172645463325252ull;
    for (auto& r : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; r = (uint32_t)(x % n_rec); }
    hipMemcpy(rows, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<40>(buf, rows, n, out, "640-B records (5 lines)");
        run<36>(buf, rows, n, out, "576-B records (4.5 lines)");
        run<34>(buf, rows, n, out, "544-B records (4.25 lines)");
    }
    return 0;
}
