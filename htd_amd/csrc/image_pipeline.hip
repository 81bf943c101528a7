// On-device data pipeline of a batch of decoded 8-bit images: one pass that reads the raw
// HWC uint8 pixels and writes the network input -- Resize (cv2 INTER_LINEAR 8U fixed point, keep-ratio sizes chosen on
// the host) -> flip -> BGR->RGB + (x - mean) * (1/std) -> zero padding to the collated batch shape -- as NHWC fp32.
// HBM-bound: 12 B written per padded output pixel, <= 12 B of (L2-resident) source bytes read per valid one.
// A thread produces 4 consecutive output pixels (48 B); a wave transposes its 3 KiB through LDS so that every
// 16 B-per-lane store instruction covers 1 KiB of consecutive bytes.
// Compiled with -ffp-contract=off: the coefficient arithmetic restates OpenCV's (double product, float rounding).
#include <algorithm>

#include "common.h"

namespace {

struct ImgMeta {          // 48 B per image, filled by the host
    int src_h, src_w, dst_h, dst_w, flip, area2, pad0, pad1;
    double scale_x, scale_y;  // 1.0 / ((double)dst / src), the doubles cv2.resize derives its tables from
};

struct NormCfg {
    float mean[3];
    double stdinv[3];
    float pad_val;
    int to_rgb;
};

__device__ __forceinline__ int sat_short(float v)
{
    const int r = (int)rintf(v);                                   // cvRound: round half to even
    return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

// source index + the two 11-bit weights of destination index d (x rule: weights reset at both borders)
__device__ __forceinline__ void coeff(int d, double scale, int src, bool clamp, int &s, int &w0, int &w1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    const float fl = floorf(f);
    f -= fl;
    s = (int)fl;
    if (clamp) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
    }
    w0 = sat_short((1.f - f) * 2048.f);
    w1 = sat_short(f * 2048.f);
}

__device__ __forceinline__ void resized_pixel(const uint8_t *__restrict__ img, const ImgMeta &m, int xr, int yr,
                                              double scale_x, int r0, int r1, int b0, int b1, int v[3])
{
    if (m.area2) {                                                  // exact 2x downscale: cv2 switches to fast INTER_AREA
        const uint8_t *p = img + ((int64_t)(2 * yr) * m.src_w + 2 * xr) * 3;
        const uint8_t *q = p + (int64_t)m.src_w * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = ((int)p[c] + (int)p[3 + c] + (int)q[c] + (int)q[3 + c] + 2) >> 2;
        return;
    }
    int sx, a0, a1;
    coeff(xr, scale_x, m.src_w, true, sx, a0, a1);
    const int sx1 = min(sx + 1, m.src_w - 1);
    const uint8_t *p0 = img + ((int64_t)r0 * m.src_w) * 3, *p1 = img + ((int64_t)r1 * m.src_w) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int s0 = (int)p0[sx * 3 + c] * a0 + (int)p0[sx1 * 3 + c] * a1;
        const int s1 = (int)p1[sx * 3 + c] * a0 + (int)p1[sx1 * 3 + c] * a1;
        v[c] = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void image_pipeline_kernel(const uint8_t *__restrict__ src,
                                                             const int64_t *__restrict__ src_off,
                                                             const ImgMeta *__restrict__ meta, NormCfg cfg,
                                                             float *__restrict__ out, int Hp, int Wp)
{
    __shared__ float4 stage[4][192];                               // per wave: 64 lanes x 3 float4, transposed on the way out
    const int b = blockIdx.y, W4 = (Wp + 3) >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < (int64_t)Hp * W4;
    const int y = (int)(t / W4), x0 = (int)(t % W4) * 4;
    const ImgMeta m = meta[b];
    const uint8_t *img = src + src_off[b];
    float px[12];
    const bool row_valid = live && y < m.dst_h;
    int r0 = 0, r1 = 0, b0 = 0, b1 = 0;
    const int yr = (m.flip & 2) ? m.dst_h - 1 - y : y;
    if (row_valid && !m.area2) {
        int sy;
        coeff(yr, m.scale_y, m.src_h, false, sy, b0, b1);
        r0 = min(max(sy, 0), m.src_h - 1);
        r1 = min(max(sy + 1, 0), m.src_h - 1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x0 + i;
        if (row_valid && x < m.dst_w) {
            const int xr = (m.flip & 1) ? m.dst_w - 1 - x : x;
            int v[3];
            resized_pixel(img, m, xr, yr, m.scale_x, r0, r1, b0, b1, v);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float u = (float)(cfg.to_rgb ? v[2 - c] : v[c]) - cfg.mean[c];
                px[i * 3 + c] = (float)((double)u * cfg.stdinv[c]);
            }
        } else {
            px[i * 3 + 0] = px[i * 3 + 1] = px[i * 3 + 2] = cfg.pad_val;
        }
    }
    float *plane = out + (int64_t)b * Hp * Wp * 3;
    if (VEC) {
        // Wp % 4 == 0: thread t owns floats [12t, 12t+12) of the image plane.  Stored straight, each 16 B store of a wave
        // would land every 48 B; through LDS each store instruction of the wave covers 1 KiB of consecutive bytes.
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        stage[wave][lane * 3 + 0] = make_float4(px[0], px[1], px[2], px[3]);
        stage[wave][lane * 3 + 1] = make_float4(px[4], px[5], px[6], px[7]);
        stage[wave][lane * 3 + 2] = make_float4(px[8], px[9], px[10], px[11]);
        __syncthreads();
        const int64_t first4 = ((int64_t)blockIdx.x * 256 + wave * 64) * 3;     // float4 index of the wave's first store
        const int64_t total4 = (int64_t)Hp * W4 * 3;
        float4 *o = reinterpret_cast<float4 *>(plane);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int64_t q = first4 + j * 64 + lane;
            if (q < total4) o[q] = stage[wave][j * 64 + lane];
        }
    } else if (live) {
        float *row = plane + (int64_t)y * Wp * 3;
        const int n = min(4, Wp - x0) * 3;
        for (int i = 0; i < n; ++i) row[(int64_t)x0 * 3 + i] = px[i];
    }
}

}  // namespace

extern "C" int htd_image_batch_pipeline(const uint8_t *src, const int64_t *src_off, const int *meta, float *out,
                                        int B, int Hp, int Wp, float mean0, float mean1, float mean2, float std0,
                                        float std1, float std2, int to_rgb, float pad_val, void *stream)
{
    HTD_REQUIRE(src && src_off && meta && out, "image_batch_pipeline: null pointer");
    HTD_REQUIRE(B > 0 && B <= 65535 && Hp > 0 && Wp > 0 && (int64_t)Hp * Wp < (1ll << 31),
                "image_batch_pipeline: bad batch shape %dx%dx%d", B, Hp, Wp);
    HTD_REQUIRE(std0 != 0.f && std1 != 0.f && std2 != 0.f, "image_batch_pipeline: std must be non-zero");
    NormCfg cfg;
    cfg.mean[0] = mean0; cfg.mean[1] = mean1; cfg.mean[2] = mean2;
    cfg.stdinv[0] = 1.0 / (double)std0; cfg.stdinv[1] = 1.0 / (double)std1; cfg.stdinv[2] = 1.0 / (double)std2;
    cfg.pad_val = pad_val;
    cfg.to_rgb = to_rgb ? 1 : 0;
    const dim3 grid((unsigned)htd::ceil_div((int64_t)Hp * ((Wp + 3) / 4), 256), (unsigned)B);
    const ImgMeta *m = reinterpret_cast<const ImgMeta *>(meta);
    if (Wp % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
        hipLaunchKernelGGL(image_pipeline_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, src_off, m, cfg, out,
                           Hp, Wp);
    else
        hipLaunchKernelGGL(image_pipeline_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, src_off, m, cfg, out,
                           Hp, Wp);
    return htd::check_launch("image_batch_pipeline");
}
