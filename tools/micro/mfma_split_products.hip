// What the socket power cap gives per ALGORITHMIC flop for the ways of carrying an fp32 product on the 16-bit matrix pipe:
//   bf16 x 6   three bf16 pieces per operand, six v_mfma_f32_32x32x16_bf16 per 32x32x16 block (conv_x3p_kernel today)
//   f16  x 3   two fp16 pieces per operand (11 + 11 bits, block-scaled), a0 b0 + a0 b1 + a1 b0
//   f16  x 4   the same with a1 b1
// each as a 64x64 wave tile whose fragments come out of LDS every step (ds_read_b128: 3 or 2 planes per operand), three
// workgroups of four waves per CU, ~0.6 s per variant so that the clock settles under the cap; and the same loops without the
// LDS reads (matrix pipe alone).  Also: does the matrix pipe keep fp16 SUBNORMAL inputs (the block-scaled scheme needs them)?
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_split_products tools/micro/mfma_split_products.hip ; run: /tmp/mfma_split_products
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool F16, int PLANES, int PRODUCTS, bool LDS>
__global__ __launch_bounds__(256, 3) void loop_kernel(const u32x4 *__restrict__ init, float *__restrict__ out, int iters)
{
    // [operand][plane][block 0..1][64 lanes] x 16 B per wave: 2 x PLANES x 2 KB
    __shared__ u32x4 lds[4 * 2 * 3 * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * 2 * 3 * 2 * 64; i += 256) lds[i] = init[(i + blockIdx.x * 17) % (4 * 2 * 3 * 2 * 64)];
    __syncthreads();
    const u32x4 *mine = lds + wave * (2 * 3 * 2 * 64) + lane;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 fa[2][3], fb[2][3];
    for (int i = 0; i < 2; ++i)
        for (int q = 0; q < 3; ++q) {
            fa[i][q] = mine[((0 * 3 + q) * 2 + i) * 64];
            fb[i][q] = mine[((1 * 3 + q) * 2 + i) * 64];
        }
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < PLANES; ++q) {
                    const volatile u32x4 *pa = mine + ((0 * 3 + q) * 2 + i) * 64, *pb = mine + ((1 * 3 + q) * 2 + i) * 64;
                    fa[i][q] = *pa;
                    fb[i][q] = *pb;
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // the pairs (piece of a, piece of b), smallest terms first as in the kernels
                constexpr int PA6[6] = {2, 0, 1, 1, 0, 0}, PB6[6] = {0, 2, 1, 0, 1, 0};
                constexpr int PA4[4] = {1, 1, 0, 0}, PB4[4] = {1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < PRODUCTS; ++t) {
                    const int qa = PRODUCTS == 6 ? PA6[t] : PA4[t + (4 - PRODUCTS)], qb = PRODUCTS == 6 ? PB6[t] : PB4[t + (4 - PRODUCTS)];
                    if constexpr (F16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][qa]),
                                                                           __builtin_bit_cast(f16x8, fb[j][qb]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][qa]),
                                                                            __builtin_bit_cast(bf16x8, fb[j][qb]), acc[i][j], 0, 0, 0);
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

__global__ void subnormal_kernel(float *out)
{
    // A = one fp16 subnormal (2^-20) at (row 0, k 0), B = 1.0 at (k 0, col 0): D[0][0] = 2^-20 when the pipe keeps subnormals
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)0.f; b[k] = (_Float16)0.f; }
    if (lane == 0) {
        a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0010);          // 16 * 2^-24 = 2^-20
        b[0] = (_Float16)1.0f;
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    // and a normal x subnormal-result product: 2^-10 * 2^-10 accumulates exactly in fp32
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)0.f; b[k] = (_Float16)0.f; }
    if (lane == 0) { a[0] = (_Float16)0.0009765625f; b[0] = (_Float16)0.0009765625f; }
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[1] = acc[0];
}

static uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s; }

template <bool F16, int PLANES, int PRODUCTS, bool LDS>
static void run(const char *name, const u32x4 *d_init, float *d_out, double clock_probe_s)
{
    const int blocks = 256 * 3;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {          // calibrate to ~clock_probe_s, report the last (longest) run
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<F16, PLANES, PRODUCTS, LDS>), dim3(blocks), dim3(256), 0, 0, d_init, d_out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep < 2) iters = (int)(iters * (clock_probe_s * 1e3 / ms) * (rep == 0 ? 0.3 : 1.0)) + 1;
    }
    const double alg = 2.0 * 64 * 64 * 16 * (double)iters * blocks * 4;
    const double mfma = (double)PRODUCTS * 4 * iters * blocks * 4;             // instructions
    const double cyc_per_simd = mfma * 32 / (256.0 * 4);                        // 8 passes x 4 cycles each, per SIMD
    printf("%-34s %8.1f ms  %7.1f algorithmic TF/s  %7.1f issued TF/s   matrix-pipe-bound clock >= %.2f GHz\n", name, ms,
           alg / ms / 1e9, alg * PRODUCTS / ms / 1e9, cyc_per_simd / (ms * 1e-3) / 1e9);
}

int main()
{
    const int n = 4 * 2 * 3 * 2 * 64;
    std::vector<uint32_t> h(n * 4);
    uint32_t s = 12345u;
    auto fill = [&](bool f16) {
        for (auto &w : h) {
            uint32_t v = 0;
            for (int half = 0; half < 2; ++half) {
                const uint32_t r = lcg(s) >> 8;
                uint32_t x;
                if (f16) x = ((r & 1) << 15) | ((10 + (r >> 1) % 8) << 10) | ((r >> 5) & 0x3ff);       // |x| in 2^-5 .. 2^2
                else x = ((r & 1) << 15) | ((122 + (r >> 1) % 8) << 7) | ((r >> 5) & 0x7f);
                v |= x << (16 * half);
            }
            w = v;
        }
    };
    u32x4 *d_init;
    float *d_out;
    hipMalloc(&d_init, n * 16);
    hipMalloc(&d_out, 256 * 3 * 256 * sizeof(float) + 64);
    subnormal_kernel<<<1, 64>>>(d_out);
    float sub[2];
    hipMemcpy(sub, d_out, sizeof(sub), hipMemcpyDeviceToHost);
    printf("fp16 subnormal input 2^-20 x 1.0 -> %g (2^-20 = %g): %s;  2^-10 x 2^-10 -> %g (exact %g)\n", sub[0], 9.5367431640625e-07,
           sub[0] == 9.5367431640625e-07f ? "kept" : "FLUSHED", sub[1], 9.5367431640625e-07);
    const double T = 0.6;
    fill(false);
    hipMemcpy(d_init, h.data(), n * 16, hipMemcpyHostToDevice);
    run<false, 3, 6, true>("bf16 x 6, fragments from LDS", d_init, d_out, T);
    run<false, 3, 6, false>("bf16 x 6, matrix pipe alone", d_init, d_out, T);
    fill(true);
    hipMemcpy(d_init, h.data(), n * 16, hipMemcpyHostToDevice);
    run<true, 2, 3, true>("f16 x 3, fragments from LDS", d_init, d_out, T);
    run<true, 2, 4, true>("f16 x 4, fragments from LDS", d_init, d_out, T);
    run<true, 2, 3, false>("f16 x 3, matrix pipe alone", d_init, d_out, T);
    run<true, 2, 4, false>("f16 x 4, matrix pipe alone", d_init, d_out, T);
    fill(false);
    hipMemcpy(d_init, h.data(), n * 16, hipMemcpyHostToDevice);
    run<false, 3, 6, true>("bf16 x 6, fragments from LDS (again)", d_init, d_out, T);
    return 0;
}
