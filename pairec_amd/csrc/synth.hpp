// synth.hpp — deterministic synthetic data generator, host+device.
//   u = splitmix64(seed ^ (row*dim + col));  value = (u>>40) * 2^-23 - 1   (uniform fp32 in [-1,1))
// so a 100 M-row table is generated on the device and any row is reproducible on the CPU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pg {

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ __forceinline__ float synth_value(uint64_t seed, uint64_t row, uint32_t col,
                                                      uint32_t dim) {
    const uint64_t u = splitmix64(seed ^ (row * (uint64_t)dim + col));
    // (u>>40) < 2^24 converts exactly; the scale and the subtraction are exact as well
    return (float)(u >> 40) * (1.0f / 8388608.0f) - 1.0f;
}

}  // namespace pg
