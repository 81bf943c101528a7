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

// Largest magnitude a launch stores, left in a device scalar for an H2 consumer of the tensor (conv_x3.hip, h2_scale): magnitudes
// order like their bit patterns, NaN is put above everything and stays.  mag_bits: one value; mag_bits4: folded into a running
// maximum; wave_mag_out: the wavefront's maximum -> *out with at most one atomic, and none when the scalar already holds as much
// (thousands of atomics on one address serialise in the L2).
__device__ __forceinline__ unsigned mag_bits(float v) { return (v != v) ? 0x7fc00000u : (__float_as_uint(v) & 0x7fffffffu); }
__device__ __forceinline__ unsigned mag_bits4(unsigned mx, float4 v)
{
    return max(max(mx, max(mag_bits(v.x), mag_bits(v.y))), max(mag_bits(v.z), mag_bits(v.w)));
}
__device__ __forceinline__ void wave_mag_out(unsigned bits, float *out)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));
    if ((threadIdx.x & 63) == 0 && bits > *reinterpret_cast<volatile unsigned *>(out))
        __hip_atomic_fetch_max(reinterpret_cast<unsigned *>(out), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace htd
