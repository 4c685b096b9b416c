// Random-row gather throughput vs row size: how many bytes does HBM move per 64-B row?  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/gather_gran.hip -o /tmp/gather_gran && /tmp/gather_gran
// One lane reads one whole row (ROWB bytes as 16-B loads) at a pseudo-random 128-B-aligned slot of a 4 GiB buffer
// (far beyond L2 + Infinity Cache); useful GB/s = rows x ROWB / time.  If 128-B rows move twice the useful bytes of
// 64-B rows in the same time, the memory side fetches 128 B per row either way (the FM kernel's 64-B field rows then
// cost 128 B of HBM traffic each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ROWB, int STRIDE>
__global__ void gather(const float4* __restrict__ buf, uint64_t slots, uint32_t per_thread, float* out) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    float acc = 0.f;
    for (uint32_t i = 0; i < per_thread; i += 8) {       // 8 rows requested together (as the FM prologue's 8 fields)
        const float4* p[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
            p[r] = buf + (x % slots) * (STRIDE / 16);
        }
        float4 v[8][ROWB / 16];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < ROWB / 16; ++j) v[r][j] = p[r][j];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < ROWB / 16; ++j) acc += v[r][j].x + v[r][j].y + v[r][j].z + v[r][j].w;
    }
    if (acc == 1.2345f) out[0] = acc;
}

// four lanes per 64-B row (16 B each): a wave instruction covers 16 rows, one 64-B segment per row
__global__ void gather4(const float4* __restrict__ buf, uint64_t slots, uint32_t per_thread, float* out) {
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t x = (tid >> 2) * 0x9E3779B97F4A7C15ull + 12345;      // the quad shares its row sequence
    const int j = threadIdx.x & 3;
    float acc = 0.f;
    for (uint32_t i = 0; i < per_thread; i += 8) {
        const float4* p[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
            p[r] = buf + (x % slots) * 4 + j;
        }
        float4 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = *p[r];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <int ROWB, int STRIDE>
static void run(const float4* buf, uint64_t bytes, float* out, const char* name) {
    const uint64_t slots = bytes / STRIDE;
    const uint32_t per_thread = 64, blocks = 256 * 16, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    gather<ROWB, STRIDE><<<blocks, threads>>>(buf, slots, per_thread, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather<ROWB, STRIDE><<<blocks, threads>>>(buf, slots, per_thread, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double rows = (double)blocks * threads * per_thread;
    printf("%-34s %8.3f ms  %7.1f G rows/s  %7.2f TB/s useful\n", name, ms, rows / ms / 1e6, rows * ROWB / ms / 1e9);
}

int main() {
    const uint64_t bytes = 4ull << 30;
    float4* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    run<64, 64>(buf, bytes, out, "64-B rows, 64-B aligned slots");
    run<64, 128>(buf, bytes, out, "64-B rows, 128-B aligned slots");
    run<128, 128>(buf, bytes, out, "128-B rows, 128-B aligned slots");
    run<32, 128>(buf, bytes, out, "32-B rows, 128-B aligned slots");
    {
        const uint64_t slots = bytes / 64;
        const uint32_t per_thread = 64, blocks = 256 * 16, threads = 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        gather4<<<blocks, threads>>>(buf, slots, per_thread, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        gather4<<<blocks, threads>>>(buf, slots, per_thread, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double rows = (double)blocks * threads * per_thread / 4.0;
        printf("%-34s %8.3f ms  %7.1f G rows/s  %7.2f TB/s useful\n", "64-B rows, 4 lanes per row", ms, rows / ms / 1e6, rows * 64 / ms / 1e9);
    }
    return 0;
}
