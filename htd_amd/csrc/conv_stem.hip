// The ResNet stem: 7x7 convolution, stride 2, padding 3, RGB (+ one zero channel) -> 64 channels, fp32 through the three-way
// bf16 split products of conv_x3.hip (six v_mfma_f32_32x32x16_bf16 per 16 k, smallest terms first, fp32 accumulate).
//
// The generic kernels want 8 input channels and spend one K slice of 8 (padded to the MFMA's 16) on every one of the 49 taps:
// 784 k per output pixel for 147 real ones.  Here the reduction runs over FILTER ROWS: for filter row ky an output pixel
// (ho, wo) meets the 7 input pixels 2 wo - 3 .. 2 wo + 3 of image row 2 ho - 3 + ky -- with 4 channels per pixel that is 28
// CONSECUTIVE values of the NHWC input, 32 with the next pixel (whose weights are zeros): two 16-k steps per filter row, 224 k
// in all.  A workgroup owns 64 consecutive output pixels of one output row (128 with -DHTD_STEM_BM=128: 246 against 216 us at
// B = 4, two resident workgroups instead of three) and all 64 output channels; it splits the 7 x 134 input pixels it needs
// ONCE into bf16 planes in LDS, in their memory order.  The A fragment of output pixel wo is the
// 16-byte window that starts at pixel 2 (wo - wo0) + 4 kk + 2 half of that image: neighbouring lanes read neighbouring
// 16-byte chunks (conflict-free), and no im2col image exists anywhere.  Weight planes ([ky][k / 8][plane][co][8]) are made
// per call by a small kernel and stream through a double-buffered 12 KB LDS tile per filter row.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#ifndef HTD_STEM_BM
#define HTD_STEM_BM 64
#endif
constexpr int ST_BM = HTD_STEM_BM;              // output pixels per workgroup: 128 (4 waves x 32 pixels x 64 channels) or 64 (2 x 2 waves)
constexpr int ST_PIX = 2 * ST_BM + 6;           // input pixels per filter row (the last one only meets zero weights)
constexpr int ST_PIXP = (ST_PIX + 7) / 8 * 8;   // row pitch in pixels
constexpr int ST_APLANE = ST_PIXP * 4;          // half-words per (filter row, plane): 4 channels per pixel
constexpr int ST_A = 7 * 3 * ST_APLANE;         // 44 352 bytes at 128 pixels
constexpr int ST_NJ = ST_BM == 128 ? 2 : 1;     // 32-column blocks per wave
constexpr int ST_BT = 4 * 3 * 64 * 8;           // half-words per filter row of weights: [k chunk][plane][co][8]
constexpr int ST_WPLANES = 7 * ST_BT;           // half-words of the whole plane image

// a = h + m + l exactly to fp32 precision, each piece a bf16 (round to nearest even); two values at a time
__device__ __forceinline__ void split3x2s(float a, float b, unsigned &h, unsigned &m, unsigned &l)
{
    union { bf16x2 v; unsigned u; } c;
    c.v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    h = c.u;
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    c.v = __builtin_convertvector(f32x2{ra, rb}, bf16x2);
    m = c.u;
    c.v = __builtin_convertvector(f32x2{ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u)}, bf16x2);
    l = c.u;
}

// w [64][7][7][4] fp32 (KRSC) -> planes [ky][k / 8][plane][co][k % 8] bf16, k = kx * 4 + c for k < 28, zeros for k = 28..31
__global__ __launch_bounds__(256) void stem7_planes_kernel(const float *__restrict__ w, unsigned short *__restrict__ planes)
{
    const int e = blockIdx.x * 256 + threadIdx.x;            // (ky, k pair, co): 7 * 16 * 64
    if (e >= 7 * 16 * 64) return;
    const int co = e % 64, kp = (e / 64) % 16, ky = e / (64 * 16);
    const int k = 2 * kp;
    float a = 0.f, b = 0.f;
    if (k < 28) {
        const float *src = w + ((co * 7 + ky) * 7 + k / 4) * 4 + (k & 3);
        a = src[0];
        b = src[1];
    }
    unsigned h, m, l;
    split3x2s(a, b, h, m, l);
    const int chunk = k / 8, off = k % 8;
    unsigned short *d = planes + ky * ST_BT + (chunk * 3 * 64 + co) * 8 + off;
    *reinterpret_cast<unsigned *>(d) = h;
    *reinterpret_cast<unsigned *>(d + 64 * 8) = m;
    *reinterpret_cast<unsigned *>(d + 2 * 64 * 8) = l;
}

struct StemParams {
    const float *x;          // [B][H][W][4]
    const unsigned short *planes;
    const float *bias;       // [64] or NULL
    float *y;                // [B][Ho][Wo][64]
    int B, H, W, Ho, Wo, segs, relu;
};

__global__ __launch_bounds__(256, (ST_BM == 128 ? 2 : 3)) void stem7_fwd_kernel(StemParams p)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[ST_A + 2 * ST_BT];
    unsigned short *la = lds, *lb = lds + ST_A;
    // XCD-aware order: the workgroups of one XCD walk a contiguous run of tiles (tiles of neighbouring output rows share
    // five of their seven input rows)
    int bid = blockIdx.x;
    {
        const int tiles = (int)gridDim.x, q = tiles / 8, r = tiles % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int seg = bid % p.segs;
    const int t2 = bid / p.segs;
    const int ho = t2 % p.Ho, b = t2 / p.Ho;
    const int wo0 = seg * ST_BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fhalf = lane >> 5;

    // ---- stage the 7 x 262 input pixels as three bf16 planes, in memory order (pixels outside the image: zeros)
    constexpr int ITEMS = 7 * ST_PIXP, PER = (ITEMS + 255) / 256;
    float4 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int it = tid + i * 256;
        const int ky = it / ST_PIXP, px = it - ky * ST_PIXP;
        const int hi = 2 * ho - 3 + ky, wi = 2 * wo0 - 3 + px;
        const bool ok = it < ITEMS && px < ST_PIX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const int64_t off = ((((int64_t)b * p.H + hi) * p.W + wi) * 4) & -(int64_t)ok;
        const float4 t = *reinterpret_cast<const float4 *>(p.x + off);
        v[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
    // the first filter row of weights
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(p.planes);
    u32x4 w0 = wsrc[tid], w1 = wsrc[tid + 256], w2 = wsrc[tid + 512];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int it = tid + i * 256;
        if (it < ITEMS) {
            const int ky = it / ST_PIXP, px = it - ky * ST_PIXP;
            unsigned h0, m0, l0, h1, m1, l1;
            split3x2s(v[i].x, v[i].y, h0, m0, l0);
            split3x2s(v[i].z, v[i].w, h1, m1, l1);
            unsigned short *d = la + (ky * 3) * ST_APLANE + px * 4;
            *reinterpret_cast<uint2 *>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(d + ST_APLANE) = make_uint2(m0, m1);
            *reinterpret_cast<uint2 *>(d + 2 * ST_APLANE) = make_uint2(l0, l1);
        }
    }
    reinterpret_cast<u32x4 *>(lb)[tid] = w0;
    reinterpret_cast<u32x4 *>(lb)[tid + 256] = w1;
    reinterpret_cast<u32x4 *>(lb)[tid + 512] = w2;
    __syncthreads();

    f32x16 acc[ST_NJ];
#pragma unroll
    for (int j = 0; j < ST_NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // 128 pixels: wave w takes output pixels wo0 + 32 w .. + 31 and all 64 channels; 64 pixels: 2 x 2 waves of 32 x 32
    const int wrow = (ST_BM == 128 ? wave : (wave & 1)) * 32, wcol = ST_BM == 128 ? 0 : (wave >> 1) * 32;
    const int a_px = 2 * (wrow + frow) + 2 * fhalf;            // first pixel of the lane's 16-byte window at kk = 0
    constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};      // smallest terms first
    for (int ky = 0; ky < 7; ++ky) {
        if (ky + 1 < 7) {
            const u32x4 *src = wsrc + (ky + 1) * (ST_BT / 8) + tid;
            w0 = src[0]; w1 = src[256]; w2 = src[512];
        }
        __builtin_amdgcn_sched_barrier(0);      // (hipcc sinks these loads behind the MFMAs and waits for them at once otherwise)
        const unsigned short *bt = lb + (ky & 1) * ST_BT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[3], fb[ST_NJ][3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                fa[q] = *reinterpret_cast<const bf16x8 *>(la + (ky * 3 + q) * ST_APLANE + (a_px + 4 * kk) * 4);
#pragma unroll
            for (int j = 0; j < ST_NJ; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    fb[j][q] = *reinterpret_cast<const bf16x8 *>(bt + (((2 * kk + fhalf) * 3 + q) * 64 + wcol + j * 32 + frow) * 8);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < ST_NJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[QA[q]], fb[j][QB[q]], acc[j], 0, 0, 0);
        }
        // (... and hoists their LDS stores, with the wait for the loads, in front of the MFMAs.  The stores are therefore made to
        // depend on the accumulators: an empty asm that "produces" one register of each staged vector from them)
        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2) : "v"(acc[0][0]), "v"(acc[ST_NJ - 1][0]));
        if (ky + 1 < 7) {
            unsigned short *nb = lb + ((ky + 1) & 1) * ST_BT;       // last read in iteration ky - 1, before its closing barrier
            reinterpret_cast<u32x4 *>(nb)[tid] = w0;
            reinterpret_cast<u32x4 *>(nb)[tid + 256] = w1;
            reinterpret_cast<u32x4 *>(nb)[tid + 512] = w2;
        }
        __syncthreads();
    }

    // ---- epilogue straight from the accumulators: D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); a
    // store instruction covers two output pixels x 32 channels = two full 128-byte lines
#pragma unroll
    for (int j = 0; j < ST_NJ; ++j) {
        const int col = wcol + j * 32 + frow;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int wo = wo0 + wrow + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            if (wo < p.Wo) {
                float t = acc[j][r] + bv;
                if (p.relu) t = fmaxf(t, 0.f);
                p.y[(((int64_t)b * p.Ho + ho) * p.Wo + wo) * 64 + col] = t;
            }
        }
    }
}

}  // namespace

extern "C" int64_t htd_conv2d_stem7_workspace_bytes(void) { return (int64_t)ST_WPLANES * 2; }

// y = act(conv7x7 stride 2 pad 3 (x, w) + bias): x [B][H][W][4] fp32 (NHWC, channel 3 zero for RGB input), w [64][7][7][4]
// fp32 (KRSC), bias [64] or NULL, y [B][Ho][Wo][64], Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1.
// workspace >= htd_conv2d_stem7_workspace_bytes().
extern "C" int htd_conv2d_stem7_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int relu,
                                    void *workspace, void *stream)
{
    HTD_REQUIRE(B > 0 && H > 0 && W > 0, "conv2d_stem7_fwd: bad sizes B=%d H=%d W=%d", B, H, W);
    HTD_REQUIRE(x && w && y && workspace, "conv2d_stem7_fwd: null pointer");
    HTD_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)workspace) & 15) == 0, "conv2d_stem7_fwd: 16-byte alignment");
    StemParams p{};
    p.x = x; p.planes = (const unsigned short *)workspace; p.bias = bias; p.y = y;
    p.B = B; p.H = H; p.W = W;
    p.Ho = (H + 6 - 7) / 2 + 1;
    p.Wo = (W + 6 - 7) / 2 + 1;
    p.segs = (int)htd::ceil_div(p.Wo, ST_BM);
    p.relu = relu;
    const int64_t tiles = (int64_t)B * p.Ho * p.segs;
    HTD_REQUIRE(tiles < (1ll << 31), "conv2d_stem7_fwd: too many tiles");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stem7_planes_kernel, dim3(7 * 16 * 64 / 256), dim3(256), 0, s, w, (unsigned short *)workspace);
    hipLaunchKernelGGL(stem7_fwd_kernel, dim3((unsigned)tiles), dim3(256), 0, s, p);
    return htd::check_launch("conv2d_stem7_fwd");
}
