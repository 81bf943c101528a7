// Deformable convolution v1 / v2 (mask == NULL => v1, the 'DCN' of configs/htd/htd_resnet101_dcn_2x_mstrain.py:142)
// for NHWC fp32 tensors: the two gather/scatter halves.  The contraction halves are the MFMA GEMM kernels of
// conv_fwd.hip / conv_wgrad.hip applied to the column matrix:
//
//   forward      columns = deform_im2col(x, offset, mask)      [M = B*Ho*Wo][K = kh*kw*C]  (tap-major, like KRSC)
//                y       = columns @ W^T                        htd_conv2d_fwd, 1x1 over M "pixels"
//   backward     gcol    = gy @ W                               htd_conv2d_bwd_data, 1x1
//                gx, goffset, gmask = deform_col2im(gcol, ...)  (this file)
//                gW      = gy^T @ columns                       htd_conv2d_bwd_weight, 1x1
//
// One wavefront per (output pixel, filter tap): the tap's sampling point is wave-uniform, lanes run along C, so the
// four bilinear neighbours are four coalesced channel-vector reads (or four 256-byte atomic adds in col2im), and
// the offset / mask gradients are wave reductions over channels.  Sampling rule = mmcv's deformable_im2col:
// zero outside (-1, H) x (-1, W), per-corner zero padding.
#include "common.h"

namespace {

struct DcnParams {
    const float *x, *offset, *mask;
    int B, H, W, C, kh, kw, stride, pad, dil, dg, Ho, Wo;
    int64_t M;
};

struct Tap {
    bool inside;
    int h_low, w_low;
    float lh, lw;
    bool ok1, ok2, ok3, ok4;
};

__device__ __forceinline__ Tap make_tap(float h, float w, int H, int W)
{
    Tap t;
    t.inside = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
    t.h_low = (int)floorf(h);
    t.w_low = (int)floorf(w);
    t.lh = h - (float)t.h_low;
    t.lw = w - (float)t.w_low;
    const int h_high = t.h_low + 1, w_high = t.w_low + 1;
    t.ok1 = t.inside && t.h_low >= 0 && t.w_low >= 0;
    t.ok2 = t.inside && t.h_low >= 0 && w_high <= W - 1;
    t.ok3 = t.inside && h_high <= H - 1 && t.w_low >= 0;
    t.ok4 = t.inside && h_high <= H - 1 && w_high <= W - 1;
    return t;
}

// offsets of tap k of deformable group g at output pixel m: offset[m][(g*taps + k)*2 + {0: dy, 1: dx}]
__device__ __forceinline__ void decode(const DcnParams &p, int64_t task, int64_t &m, int &k, int &b, int &ho, int &wo)
{
    const int taps = p.kh * p.kw;
    k = (int)(task % taps);
    m = task / taps;
    wo = (int)(m % p.Wo);
    const int64_t t = m / p.Wo;
    ho = (int)(t % p.Ho);
    b = (int)(t / p.Ho);
}

__global__ __launch_bounds__(256) void deform_im2col_kernel(DcnParams p, float *__restrict__ col)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int taps = p.kh * p.kw;
    if (task >= p.M * taps) return;
    int64_t m; int k, b, ho, wo;
    decode(p, task, m, k, b, ho, wo);
    const int ky = k / p.kw, kx = k % p.kw;
    const int cpg = p.C / p.dg;
    const float *img = p.x + (int64_t)b * p.H * p.W * p.C;
    float *out = col + (m * taps + k) * p.C;
    for (int c = lane * 4; c < p.C; c += 256) {
        const int g = c / cpg;
        const int64_t ob = m * (int64_t)(p.dg * taps * 2) + (int64_t)(g * taps + k) * 2;
        const float hs = (float)(ho * p.stride - p.pad + ky * p.dil) + p.offset[ob];
        const float ws = (float)(wo * p.stride - p.pad + kx * p.dil) + p.offset[ob + 1];
        const Tap t = make_tap(hs, ws, p.H, p.W);
        const float hh = 1.f - t.lh, hw = 1.f - t.lw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        auto acc = [&](bool ok, int y, int x, float wgt) {
            if (!ok) return;
            const float4 s = *reinterpret_cast<const float4 *>(img + ((int64_t)y * p.W + x) * p.C + c);
            v.x += wgt * s.x; v.y += wgt * s.y; v.z += wgt * s.z; v.w += wgt * s.w;
        };
        acc(t.ok1, t.h_low, t.w_low, hh * hw);
        acc(t.ok2, t.h_low, t.w_low + 1, hh * t.lw);
        acc(t.ok3, t.h_low + 1, t.w_low, t.lh * hw);
        acc(t.ok4, t.h_low + 1, t.w_low + 1, t.lh * t.lw);
        if (p.mask) {
            const float mk = p.mask[m * (int64_t)(p.dg * taps) + g * taps + k];
            v.x *= mk; v.y *= mk; v.z *= mk; v.w *= mk;
        }
        *reinterpret_cast<float4 *>(out + c) = v;
    }
}

// gx must be zero-initialised; goffset [M][dg*taps*2], gmask [M][dg*taps] are written in full.
__global__ __launch_bounds__(256) void deform_col2im_kernel(DcnParams p, const float *__restrict__ gcol,
                                                            float *__restrict__ gx, float *__restrict__ goffset,
                                                            float *__restrict__ gmask)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int taps = p.kh * p.kw;
    if (task >= p.M * taps) return;
    int64_t m; int k, b, ho, wo;
    decode(p, task, m, k, b, ho, wo);
    const int ky = k / p.kw, kx = k % p.kw;
    const int cpg = p.C / p.dg;
    const float *img = p.x + (int64_t)b * p.H * p.W * p.C;
    float *gimg = gx ? gx + (int64_t)b * p.H * p.W * p.C : nullptr;
    const float *gc = gcol + (m * taps + k) * p.C;
    for (int g = 0; g < p.dg; ++g) {
        const int64_t ob = m * (int64_t)(p.dg * taps * 2) + (int64_t)(g * taps + k) * 2;
        const float hs = (float)(ho * p.stride - p.pad + ky * p.dil) + p.offset[ob];
        const float ws = (float)(wo * p.stride - p.pad + kx * p.dil) + p.offset[ob + 1];
        const Tap t = make_tap(hs, ws, p.H, p.W);
        const float hh = 1.f - t.lh, hw = 1.f - t.lw;
        const float mk = p.mask ? p.mask[m * (int64_t)(p.dg * taps) + g * taps + k] : 1.f;
        float s_dy = 0.f, s_dx = 0.f, s_mk = 0.f;
        for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
            const float gv = gc[c];
            const float v1 = t.ok1 ? img[((int64_t)t.h_low * p.W + t.w_low) * p.C + c] : 0.f;
            const float v2 = t.ok2 ? img[((int64_t)t.h_low * p.W + t.w_low + 1) * p.C + c] : 0.f;
            const float v3 = t.ok3 ? img[((int64_t)(t.h_low + 1) * p.W + t.w_low) * p.C + c] : 0.f;
            const float v4 = t.ok4 ? img[((int64_t)(t.h_low + 1) * p.W + t.w_low + 1) * p.C + c] : 0.f;
            // d(val)/dh = hw*(v3-v1) + lw*(v4-v2) ; d(val)/dw = hh*(v2-v1) + lh*(v4-v3)
            s_dy += gv * mk * (hw * (v3 - v1) + t.lw * (v4 - v2));
            s_dx += gv * mk * (hh * (v2 - v1) + t.lh * (v4 - v3));
            s_mk += gv * (hh * hw * v1 + hh * t.lw * v2 + t.lh * hw * v3 + t.lh * t.lw * v4);
            if (gimg) {
                const float gm = gv * mk;
                if (t.ok1) atomicAdd(gimg + ((int64_t)t.h_low * p.W + t.w_low) * p.C + c, gm * hh * hw);
                if (t.ok2) atomicAdd(gimg + ((int64_t)t.h_low * p.W + t.w_low + 1) * p.C + c, gm * hh * t.lw);
                if (t.ok3) atomicAdd(gimg + ((int64_t)(t.h_low + 1) * p.W + t.w_low) * p.C + c, gm * t.lh * hw);
                if (t.ok4) atomicAdd(gimg + ((int64_t)(t.h_low + 1) * p.W + t.w_low + 1) * p.C + c, gm * t.lh * t.lw);
            }
        }
        s_dy = htd::wave_sum(s_dy);
        s_dx = htd::wave_sum(s_dx);
        s_mk = htd::wave_sum(s_mk);
        if (lane == 0) {
            if (goffset) { goffset[ob] = s_dy; goffset[ob + 1] = s_dx; }
            if (gmask) gmask[m * (int64_t)(p.dg * taps) + g * taps + k] = s_mk;
        }
    }
}

int fill(DcnParams &p, const float *x, const float *offset, const float *mask, int B, int H, int W, int C, int kh,
         int kw, int stride, int pad, int dil, int dg)
{
    HTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && dil > 0 && pad >= 0 && dg > 0,
                "deform_conv: bad sizes");
    HTD_REQUIRE(C % dg == 0 && (C / dg) % 4 == 0, "deform_conv: channels per deformable group (%d/%d) must be a multiple of 4",
                C, dg);
    HTD_REQUIRE(x && offset, "deform_conv: null pointer");
    p.x = x; p.offset = offset; p.mask = mask;
    p.B = B; p.H = H; p.W = W; p.C = C; p.kh = kh; p.kw = kw; p.stride = stride; p.pad = pad; p.dil = dil; p.dg = dg;
    p.Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    p.Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    HTD_REQUIRE(p.Ho > 0 && p.Wo > 0, "deform_conv: empty output");
    p.M = (int64_t)B * p.Ho * p.Wo;
    return HTD_OK;
}

}  // namespace

extern "C" int64_t htd_deform_columns_bytes(int B, int H, int W, int C, int kh, int kw, int stride, int pad, int dil)
{
    const int64_t Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    const int64_t Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    return (int64_t)B * Ho * Wo * kh * kw * C * 4;
}

extern "C" int htd_deform_im2col(const float *x, const float *offset, const float *mask, float *columns, int B, int H,
                                 int W, int C, int kh, int kw, int stride, int pad, int dil, int deform_groups,
                                 void *stream)
{
    DcnParams p{};
    const int st = fill(p, x, offset, mask, B, H, W, C, kh, kw, stride, pad, dil, deform_groups);
    if (st) return st;
    HTD_REQUIRE(columns, "deform_im2col: null columns");
    const int64_t tasks = p.M * kh * kw;
    const int64_t blocks = htd::ceil_div(tasks, 4);
    HTD_REQUIRE(blocks < (1ll << 31), "deform_im2col: too many tasks");
    hipLaunchKernelGGL(deform_im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, columns);
    return htd::check_launch("deform_im2col");
}

extern "C" int htd_deform_col2im(const float *x, const float *offset, const float *mask, const float *gcolumns,
                                 float *gx, float *goffset, float *gmask, int B, int H, int W, int C, int kh, int kw,
                                 int stride, int pad, int dil, int deform_groups, void *stream)
{
    DcnParams p{};
    const int st = fill(p, x, offset, mask, B, H, W, C, kh, kw, stride, pad, dil, deform_groups);
    if (st) return st;
    HTD_REQUIRE(gcolumns, "deform_col2im: null gradient columns");
    const int64_t tasks = p.M * kh * kw;
    const int64_t blocks = htd::ceil_div(tasks, 4);
    HTD_REQUIRE(blocks < (1ll << 31), "deform_col2im: too many tasks");
    hipLaunchKernelGGL(deform_col2im_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, gcolumns, gx,
                       goffset, gmask);
    return htd::check_launch("deform_col2im");
}
