// ds_read_b128 cost of the operand-fragment address patterns of conv_fwd.hip (lane -> row = lane & 31, 16-B half = lane >> 5).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_read_b128 tools/micro/lds_read_b128.hip ;  run: /tmp/lds_read_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(const int *lane_off, float *out, int iters, int extra)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = (float)i;
    __syncthreads();
    const unsigned off = (unsigned)lane_off[lane];
    const unsigned base = (unsigned)(size_t)lds;      // LDS byte address of the dynamic segment
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            f4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base + off + u * extra));
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            acc.x += __builtin_bit_cast(float, it);      // results are not consumed: the loop measures issue + LDS cycles
            (void)v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
    struct Pat { const char *name; int (*f)(int); int extra; };
    static Pat pats[] = {
        {"lane-linear 16 B (conflict-free reference)", [](int l) { return l * 16; }, 1024},
        {"row stride 208 B (BK=32 planes + 16 B pad), half at +16", [](int l) { return (l & 31) * 208 + (l >> 5) * 16; }, 64},
        {"row stride 112 B (BK=16 planes + 16 B pad), half at +16", [](int l) { return (l & 31) * 112 + (l >> 5) * 16; }, 32},
        {"packed 96 B rows, octet shift 16 B, block 1552 B", [](int l) { int r = l & 31; return (r >> 4) * 1552 + (r & 15) * 96 + ((r >> 3) & 1) * 16 + (l >> 5) * 16; }, 32},
        {"row stride 208 B, both halves in one row group (l>>5 -> +6656 B = row+32)", [](int l) { return (l & 31) * 208 + (l >> 5) * 6656; }, 64},
        {"row stride 144 B (fp32 [row][32+4]), half at +16", [](int l) { return (l & 31) * 144 + (l >> 5) * 16; }, 32},
        {"row stride 80 B, half at +16", [](int l) { return (l & 31) * 80 + (l >> 5) * 16; }, 32},
        {"row stride 272 B, half at +16", [](int l) { return (l & 31) * 272 + (l >> 5) * 16; }, 32},
        {"row stride 528 B, half at +16", [](int l) { return (l & 31) * 528 + (l >> 5) * 16; }, 32},
    };
    int *d_off; float *d_out;
    hipMalloc(&d_off, 64 * sizeof(int));
    const int wgs_per_cu = 4, blocks = 256 * wgs_per_cu, iters = 2000;      // 16 waves per CU
    hipMalloc(&d_out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &p : pats) {
        int h[64];
        for (int l = 0; l < 64; ++l) h[l] = p.f(l);
        hipMemcpy(d_off, h, sizeof(h), hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 32768, 0, d_off, d_out, iters, p.extra);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // one workgroup of 4 waves per CU: reads per CU = 4 waves * iters * 12
        const double reads = 4.0 * wgs_per_cu * iters * 12;
        printf("%-78s %7.3f ms  %6.2f ns per wave-read per CU\n", p.name, ms, ms * 1e6 / reads);
    }
    return 0;
}
