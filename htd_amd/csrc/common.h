// Shared helpers for the libhtd_amd.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/htd_amd.h"

namespace htd {

void set_error(const char *fmt, ...);
int conv_math();      // 1: fp32 products through three-way bf16 splits (htd_conv2d_set_math), 0: fp32-input MFMA

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return HTD_ERR_LAUNCH;
    }
    return HTD_OK;
}

#define HTD_REQUIRE(cond, ...)               \
    do {                                     \
        if (!(cond)) {                       \
            htd::set_error(__VA_ARGS__);     \
            return HTD_ERR_ARG;              \
        }                                    \
    } while (0)

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace htd
