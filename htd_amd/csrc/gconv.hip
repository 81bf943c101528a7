// Grouped 3x3 convolutions of the ResNeXt bottlenecks (groups = 64, 4..32 channels per group;
// backbones/resnext.py:33-84) on NHWC fp32 activations: forward, data gradient, weight gradient, and the same three
// over a pre-gathered column buffer [M][taps][C] for the grouped deformable conv2 of the X101-DCN config.
//
// These layers carry 2.25*Cg flop per activation byte -- HBM-bound at Cg = 4/8, balanced at 16/32 -- so there is no
// LDS-tiled implicit GEMM here.  Channels are cut into 16-wide SLABS; one wave owns 64 output pixels x one output slab
// and walks (input slab, tap) with v_mfma_f32_16x16x4_f32, both operands loaded straight from global memory as one
// float4 per lane: lane (r = l&15, j = l>>4) reads channels 4j..4j+3 of pixel r (A) or of output channel r (B), and
// MFMA step s = 0..3 consumes component s -- the k index of step s, lane-quad j is channel 4j+s on both sides, so no
// lane movement is needed.  A 16-channel slab of a pixel is one 64 B segment; the 9 taps re-read it from L1/L2.
// Groups narrower than a slab (Cg = 4, 8) share one with block-diagonal zeros in the packed weights (the MFMA pipe is
// idle most of the time on those layers anyway); groups wider than a slab (Cg = 32) sum over Cg/16 input slabs.
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { G_CONV = 0, G_COLS = 1, G_DGRAD = 2, G_DGRAD_COLS = 3 };

struct GGeo {
    int B, H, W, C, Ho, Wo, kh, kw, stride, pad, dil, cg, islabs;   // islabs = input slabs per output slab
    int64_t M;                                                      // output rows of the launch
};

__device__ __forceinline__ int in_slab_of(const GGeo &g, int os, int is)
{
    return g.islabs == 1 ? os : (os / g.islabs) * g.islabs + is;
}

// ----------------------------------------------------------------------------------------------- weight packing
// wp[((os*islabs + is)*taps + tap)*64 + l] (float4), l = n + 16 j, component s:
//   forward  : w[out = 16 os + n][tap][in  = 16 in_slab + 4j + s]      (zero across groups)
//   transpose: w[out = 16 in_slab + 4j + s][tap][in = 16 os + n]       (the data gradient's operand)
// with w the grouped KRSC tensor [C][taps][cg].
__global__ void gconv_pack_kernel(const float *__restrict__ w, float *__restrict__ wp, GGeo g, int taps, int transpose,
                                  int64_t total)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(i & 3), l = (int)((i >> 2) & 63);
        int64_t r = i >> 8;
        const int tap = (int)(r % taps);
        r /= taps;
        const int is = (int)(r % g.islabs), os = (int)(r / g.islabs);
        const int a = 16 * os + (l & 15), c = 16 * in_slab_of(g, os, is) + 4 * (l >> 4) + s;
        const int out = transpose ? c : a, in = transpose ? a : c;
        float v = 0.f;
        if (out / g.cg == in / g.cg) v = w[((int64_t)out * taps + tap) * g.cg + (in % g.cg)];
        wp[i] = v;
    }
}

// ----------------------------------------------------------------------------------------------- forward / dgrad
// Element offset (into the [rows][C] source) of the pixel feeding an output pixel through tap (dy, dx), or -1.
// `oy`, `ox`: the output pixel (G_CONV) or the input-space pixel (G_DGRAD); `pb` = image index.
template <int MODE>
__device__ __forceinline__ int64_t tap_offset(const GGeo &g, int64_t m, int pb, int oy, int ox, int dy, int dx, int tap,
                                              int taps)
{
    if (MODE == G_COLS) return (m * taps + tap) * g.C;
    if (MODE == G_DGRAD_COLS) return m * g.C;
    if (MODE == G_CONV) {
        const int iy = oy * g.stride - g.pad + dy * g.dil, ix = ox * g.stride - g.pad + dx * g.dil;
        if ((unsigned)iy >= (unsigned)g.H || (unsigned)ix >= (unsigned)g.W) return -1;
        return (((int64_t)pb * g.H + iy) * g.W + ix) * g.C;
    }
    int ty = oy + g.pad - dy * g.dil, tx = ox + g.pad - dx * g.dil;
    if (ty < 0 || tx < 0) return -1;
    if (g.stride == 2) {
        if ((ty | tx) & 1) return -1;
        ty >>= 1;
        tx >>= 1;
    } else if (g.stride != 1) {
        if (ty % g.stride || tx % g.stride) return -1;
        ty /= g.stride;
        tx /= g.stride;
    }
    if (ty >= g.Ho || tx >= g.Wo) return -1;
    return (((int64_t)pb * g.Ho + ty) * g.Wo + tx) * g.C;
}

// KW3: the kernel is 3 taps wide (every layer of the path), so (dy, dx) of an unrolled tap are constants.
template <int MODE, bool KW3>
__global__ __launch_bounds__(256) void gconv_kernel(const float *__restrict__ x, const f32x4 *__restrict__ wp,
                                                    const float *__restrict__ bias, float *__restrict__ y, GGeo g, int relu)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slabs = g.C >> 4, slab_blocks = (slabs + 3) >> 2;
    const int os0 = (int)(blockIdx.x % slab_blocks) * 4;               // 4 adjacent slabs per block: 256 B per pixel
    const int os = os0 + wave;
    const int64_t m0 = (int64_t)(blockIdx.x / slab_blocks) * 64;
    const bool active = os < slabs;                                    // (a block's last waves idle when C % 64 != 0)
    const int r = lane & 15, j = lane >> 4;
    const int taps = g.kh * g.kw;
    const int only_tap = (MODE == G_DGRAD_COLS) ? (int)blockIdx.y : -1;
    int pb[4], py[4], px[4];
    bool live[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t m = m0 + t * 16 + r;
        live[t] = m < g.M;
        pb[t] = py[t] = px[t] = 0;
        if (live[t] && (MODE == G_CONV || MODE == G_DGRAD)) {
            const unsigned w_ = (MODE == G_CONV) ? g.Wo : g.W, h_ = (MODE == G_CONV) ? g.Ho : g.H;
            const unsigned mu = (unsigned)m, q = mu / w_;              // M < 2^31 (checked by the launcher)
            px[t] = (int)(mu - q * w_);
            pb[t] = (int)(q / h_);
            py[t] = (int)(q - (unsigned)pb[t] * h_);
        }
    }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int is = 0; active && is < g.islabs; ++is) {
        const float *xs = x + 16 * in_slab_of(g, os, is) + 4 * j;
        const f32x4 *wq = wp + ((int64_t)(os * g.islabs + is) * taps) * 64 + lane;
        auto one_tap = [&](int tap, int dy, int dx) {
            const f32x4 bw = wq[(int64_t)tap * 64];
            f32x4 a[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (live[t]) {
                    const int64_t off = tap_offset<MODE>(g, m0 + t * 16 + r, pb[t], py[t], px[t], dy, dx, tap, taps);
                    if (off >= 0) a[t] = *reinterpret_cast<const f32x4 *>(xs + off);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, bw.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, bw.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, bw.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, bw.w, acc[t], 0, 0, 0);
            }
        };
        if (MODE == G_DGRAD_COLS) {
            one_tap(only_tap, 0, 0);
        } else if (KW3) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                if (tap < taps) one_tap(tap, tap / 3, tap % 3);
        } else {
            for (int tap = 0; tap < taps; ++tap) one_tap(tap, tap / g.kw, tap % g.kw);
        }
    }
    // D: lane holds rows 4j + i (pixels), column r (channel) of each 16x16 tile.  The block's 64 pixels x 64 channels go
    // through LDS so that a store instruction writes 16 B per lane along whole 256 B pixel rows (straight from the
    // accumulators it would be 4 B per lane in 64 B pieces)
    __shared__ float stage[64][68];
    if (active) {
        const float bv = bias ? bias[16 * os + r] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[t][i] + bv;
                if (relu) v = fmaxf(v, 0.f);
                stage[t * 16 + 4 * j + i][wave * 16 + r] = v;
            }
    }
    __syncthreads();
    const int q = threadIdx.x & 15, ch = os0 * 16 + q * 4;
    if (ch < g.C) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int pl = (threadIdx.x >> 4) + pass * 16;
            const int64_t m = m0 + pl;
            if (m >= g.M) continue;
            const int64_t orow = (MODE == G_DGRAD_COLS) ? m * taps + only_tap : m;
            *reinterpret_cast<f32x4 *>(y + orow * g.C + ch) = *reinterpret_cast<const f32x4 *>(&stage[pl][q * 4]);
        }
    }
}

// ----------------------------------------------------------------------------------------------- weight gradient
// One wave: output slab os x input slab is x pixel chunk; D[n][c] per tap (taps <= 9 tiles of 16x16) summed over the
// chunk's pixels, 4 pixels per MFMA: A[n][k] = gy[m+k][16 os + n], B[k][c] = x[src(m+k, tap)][16 in_slab + c].
// Partial tiles go to ws[chunk][os*islabs + is][tap][n][c]; gconv_wgrad_reduce_kernel sums the chunks and drops the
// cross-group zeros.
template <int MODE, bool KW3>
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                          float *__restrict__ ws, GGeo g, int64_t chunk_rows)
{
    constexpr int MAXT = 9;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pairs = (g.C >> 4) * g.islabs;
    const int pair = (int)blockIdx.x * 4 + wave;
    if (pair >= pairs) return;
    const int os = pair / g.islabs, is = pair - os * g.islabs;
    const int taps = g.kh * g.kw;
    const int r = lane & 15, k = lane >> 4;
    const int64_t begin = (int64_t)blockIdx.y * chunk_rows, end = std::min<int64_t>(g.M, begin + chunk_rows);
    const float *gyp = gy + 16 * os + r;
    const float *xp = x + 16 * in_slab_of(g, os, is) + r;
    f32x4 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this lane's pixel m = begin + k, advanced by 4 per step; (b, oy, ox) follow incrementally
    int b = 0, oy = 0, ox = 0;
    if (MODE == G_CONV && begin + k < g.M) {
        const unsigned mu = (unsigned)(begin + k), q = mu / (unsigned)g.Wo;
        ox = (int)(mu - q * (unsigned)g.Wo);
        b = (int)(q / (unsigned)g.Ho);
        oy = (int)(q - (unsigned)b * (unsigned)g.Ho);
    }
    for (int64_t m = begin + k; m - k < end; m += 4) {
        const bool ok = m < end;
        const float a = ok ? gyp[m * g.C] : 0.f;
        float v[MAXT];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            v[t] = 0.f;
            if (t < taps && ok) {
                const int dy = KW3 ? t / 3 : t / g.kw, dx = KW3 ? t % 3 : t % g.kw;
                const int64_t off = tap_offset<MODE>(g, m, b, oy, ox, dy, dx, t, taps);
                if (off >= 0) v[t] = xp[off];
            }
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < taps) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[t], acc[t], 0, 0, 0);
        if (MODE == G_CONV) {
            ox += 4;
            while (ox >= g.Wo) {
                ox -= g.Wo;
                if (++oy == g.Ho) { oy = 0; ++b; }
            }
        }
    }
    float *o = ws + (((int64_t)blockIdx.y * pairs + pair) * taps) * 256;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < taps) {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[(int64_t)t * 256 + (4 * k + i) * 16 + r] = acc[t][i];
        }
    }
}

// gw[out][tap][ci] (grouped KRSC, ci < cg) = sum over chunks of the slab tile holding (out, in = group*cg + ci)
__global__ void gconv_wgrad_reduce_kernel(const float *__restrict__ ws, float *__restrict__ gw, GGeo g, int taps,
                                          int chunks, int64_t total)
{
    const int pairs = (g.C >> 4) * g.islabs;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % g.cg);
        int64_t q = i / g.cg;
        const int tap = (int)(q % taps), out = (int)(q / taps);
        const int in = (out / g.cg) * g.cg + ci;
        const int os = out >> 4, islab = in >> 4;
        const int is = g.islabs == 1 ? 0 : islab - (os / g.islabs) * g.islabs;
        const int64_t off = (((int64_t)(os * g.islabs + is)) * taps + tap) * 256 + (out & 15) * 16 + (in & 15);
        float s = 0.f;
        for (int c = 0; c < chunks; ++c) s += ws[(int64_t)c * pairs * taps * 256 + off];
        gw[i] = s;
    }
}

int fill_geo(GGeo &g, const char *what, int B, int H, int W, int C, int groups, int kh, int kw, int stride, int pad, int dil)
{
    HTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && groups > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0 && dil > 0,
                "%s: bad sizes", what);
    HTD_REQUIRE(C % 16 == 0 && C % groups == 0, "%s: C=%d must be a multiple of 16 and of groups=%d", what, C, groups);
    const int cg = C / groups;
    HTD_REQUIRE(cg % 4 == 0 && (16 % cg == 0 || cg % 16 == 0), "%s: %d channels per group unsupported (4, 8, 16, 32, ...)",
                what, cg);
    HTD_REQUIRE(kh * kw <= 9, "%s: at most 9 taps", what);
    g.B = B; g.H = H; g.W = W; g.C = C; g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad; g.dil = dil;
    g.Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1;
    g.Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
    HTD_REQUIRE(g.Ho > 0 && g.Wo > 0, "%s: empty output", what);
    g.cg = cg;
    g.islabs = cg <= 16 ? 1 : cg / 16;
    g.M = 0;
    return HTD_OK;
}

int wgrad_chunks(const GGeo &g)
{
    const int pairs = (g.C >> 4) * g.islabs;
    const int64_t blocks = htd::ceil_div(pairs, 4);
    int64_t chunks = std::max<int64_t>(1, htd::ceil_div(2048, blocks));        // >= 8 workgroups per CU in flight
    chunks = std::min<int64_t>(chunks, htd::ceil_div(g.M, 256));               // >= 256 pixels per chunk
    return (int)std::max<int64_t>(chunks, 1);
}

}  // namespace

extern "C" int64_t htd_gconv2d_packed_floats(int C, int groups, int kh, int kw)
{
    if (C <= 0 || groups <= 0 || C % groups || C % 16) return -1;
    const int cg = C / groups;
    const int islabs = cg <= 16 ? 1 : cg / 16;
    return (int64_t)(C / 16) * islabs * kh * kw * 256;
}

extern "C" int htd_gconv2d_pack_weights(const float *w, float *wp, int C, int groups, int kh, int kw, int transpose,
                                        void *stream)
{
    GGeo g;
    if (int e = fill_geo(g, "gconv2d_pack_weights", 1, 8, 8, C, groups, kh, kw, 1, 1, 1)) return e;
    HTD_REQUIRE(w && wp, "gconv2d_pack_weights: null pointer");
    const int64_t total = htd_gconv2d_packed_floats(C, groups, kh, kw);
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(gconv_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp, g, kh * kw, transpose, total);
    return htd::check_launch("gconv2d_pack_weights");
}

template <int MODE>
static int launch_gconv(const char *what, const float *x, const float *wp, const float *bias, float *y, GGeo &g, int relu,
                        void *stream)
{
    const int slab_blocks = ((g.C >> 4) + 3) >> 2;
    const int64_t blocks = htd::ceil_div(g.M, 64) * slab_blocks;
    HTD_REQUIRE(blocks < (1ll << 31), "%s: launch too large", what);
    const dim3 grid((unsigned)blocks, MODE == G_DGRAD_COLS ? (unsigned)(g.kh * g.kw) : 1u);
    HTD_REQUIRE(g.M < (1ll << 31) && (int64_t)g.B * g.H * g.W < (1ll << 31), "%s: more than 2^31 pixels", what);
    if (g.kw == 3)
        hipLaunchKernelGGL((gconv_kernel<MODE, true>), grid, dim3(256), 0, (hipStream_t)stream, x,
                           reinterpret_cast<const f32x4 *>(wp), bias, y, g, relu);
    else
        hipLaunchKernelGGL((gconv_kernel<MODE, false>), grid, dim3(256), 0, (hipStream_t)stream, x,
                           reinterpret_cast<const f32x4 *>(wp), bias, y, g, relu);
    return htd::check_launch(what);
}

extern "C" int htd_gconv2d_fwd(const float *x, const float *wp, const float *bias, float *y, int B, int H, int W, int C,
                               int groups, int kh, int kw, int stride, int pad, int dil, int relu, int cols, void *stream)
{
    GGeo g;
    if (int e = fill_geo(g, "gconv2d_fwd", B, H, W, C, groups, kh, kw, stride, pad, dil)) return e;
    HTD_REQUIRE(x && wp && y, "gconv2d_fwd: null pointer");
    g.M = (int64_t)B * g.Ho * g.Wo;
    return cols ? launch_gconv<G_COLS>("gconv2d_fwd", x, wp, bias, y, g, relu, stream)
                : launch_gconv<G_CONV>("gconv2d_fwd", x, wp, bias, y, g, relu, stream);
}

extern "C" int htd_gconv2d_bwd_data(const float *gy, const float *wpT, float *gx, int B, int H, int W, int C, int groups,
                                    int kh, int kw, int stride, int pad, int dil, int cols, void *stream)
{
    GGeo g;
    if (int e = fill_geo(g, "gconv2d_bwd_data", B, H, W, C, groups, kh, kw, stride, pad, dil)) return e;
    HTD_REQUIRE(gy && wpT && gx, "gconv2d_bwd_data: null pointer");
    if (cols) {
        g.M = (int64_t)B * g.Ho * g.Wo;
        return launch_gconv<G_DGRAD_COLS>("gconv2d_bwd_data", gy, wpT, nullptr, gx, g, 0, stream);
    }
    g.M = (int64_t)B * H * W;
    return launch_gconv<G_DGRAD>("gconv2d_bwd_data", gy, wpT, nullptr, gx, g, 0, stream);
}

extern "C" int64_t htd_gconv2d_wgrad_workspace_bytes(int B, int H, int W, int C, int groups, int kh, int kw, int stride,
                                                     int pad, int dil)
{
    GGeo g;
    if (fill_geo(g, "gconv2d_wgrad_workspace_bytes", B, H, W, C, groups, kh, kw, stride, pad, dil)) return -1;
    g.M = (int64_t)B * g.Ho * g.Wo;
    return (int64_t)wgrad_chunks(g) * (C / 16) * g.islabs * kh * kw * 256 * 4;
}

extern "C" int htd_gconv2d_bwd_weight(const float *x, const float *gy, float *gw, int B, int H, int W, int C, int groups,
                                      int kh, int kw, int stride, int pad, int dil, int cols, void *workspace, void *stream)
{
    GGeo g;
    if (int e = fill_geo(g, "gconv2d_bwd_weight", B, H, W, C, groups, kh, kw, stride, pad, dil)) return e;
    HTD_REQUIRE(x && gy && gw && workspace, "gconv2d_bwd_weight: null pointer");
    g.M = (int64_t)B * g.Ho * g.Wo;
    const int chunks = wgrad_chunks(g);
    const int64_t chunk_rows = htd::ceil_div(htd::ceil_div(g.M, chunks), 4) * 4;
    const int pairs = (C / 16) * g.islabs;
    const dim3 grid((unsigned)htd::ceil_div(pairs, 4), (unsigned)chunks);
    float *ws = static_cast<float *>(workspace);
    HTD_REQUIRE(g.M < (1ll << 31), "gconv2d_bwd_weight: more than 2^31 pixels");
    if (cols && kw == 3)
        hipLaunchKernelGGL((gconv_wgrad_kernel<G_COLS, true>), grid, dim3(256), 0, (hipStream_t)stream, x, gy, ws, g, chunk_rows);
    else if (cols)
        hipLaunchKernelGGL((gconv_wgrad_kernel<G_COLS, false>), grid, dim3(256), 0, (hipStream_t)stream, x, gy, ws, g, chunk_rows);
    else if (kw == 3)
        hipLaunchKernelGGL((gconv_wgrad_kernel<G_CONV, true>), grid, dim3(256), 0, (hipStream_t)stream, x, gy, ws, g, chunk_rows);
    else
        hipLaunchKernelGGL((gconv_wgrad_kernel<G_CONV, false>), grid, dim3(256), 0, (hipStream_t)stream, x, gy, ws, g, chunk_rows);
    if (int e = htd::check_launch("gconv2d_bwd_weight")) return e;
    const int64_t total = (int64_t)C * kh * kw * g.cg;
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(gconv_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ws, gw, g, kh * kw, chunks,
                       total);
    return htd::check_launch("gconv2d_bwd_weight(reduce)");
}
