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
#include <stdlib.h>

#include "common.h"

namespace {

// Element types of x / columns / gradient columns: float, or bf16 carried as its 16-bit pattern (the bf16 mode of the
// detector keeps activations in bf16; sampling arithmetic, offsets and every accumulation stay fp32).
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ bf16_t f2bf(float f)
{
    unsigned u = __float_as_uint(f);
    if (f != f) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);                                // round to nearest even
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float ld1(const float *p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t *p) { return bf2f(*p); }
__device__ __forceinline__ float4 ld4v(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld4v(const bf16_t *p)
{
    const uint2 r = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4v(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4v(bf16_t *p, float4 v)
{
    uint2 r;
    r.x = (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16);
    r.y = (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16);
    *reinterpret_cast<uint2 *>(p) = r;
}

struct DcnParams {
    const void *x;
    const float *offset, *mask;
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

template <typename T>
__global__ __launch_bounds__(256) void deform_im2col_kernel(DcnParams p, T *__restrict__ col)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int taps = p.kh * p.kw;
    if (task >= p.M * taps) return;
    int64_t m; int k, b, ho, wo;
    decode(p, task, m, k, b, ho, wo);
    const int ky = k / p.kw, kx = k % p.kw;
    const int cpg = p.C / p.dg;
    const T *img = static_cast<const T *>(p.x) + (int64_t)b * p.H * p.W * p.C;
    T *out = col + (m * taps + k) * p.C;
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
            const float4 s = ld4v(img + ((int64_t)y * p.W + x) * p.C + c);
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
        st4v(out + c, v);
    }
}

// gx must be zero-initialised; goffset [M][dg*taps*2], gmask [M][dg*taps] are written in full.
template <typename T>
__global__ __launch_bounds__(256) void deform_col2im_kernel(DcnParams p, const T *__restrict__ gcol,
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
    const T *img = static_cast<const T *>(p.x) + (int64_t)b * p.H * p.W * p.C;
    float *gimg = gx ? gx + (int64_t)b * p.H * p.W * p.C : nullptr;
    const T *gc = gcol + (m * taps + k) * p.C;
    for (int g = 0; g < p.dg; ++g) {
        const int64_t ob = m * (int64_t)(p.dg * taps * 2) + (int64_t)(g * taps + k) * 2;
        const float hs = (float)(ho * p.stride - p.pad + ky * p.dil) + p.offset[ob];
        const float ws = (float)(wo * p.stride - p.pad + kx * p.dil) + p.offset[ob + 1];
        const Tap t = make_tap(hs, ws, p.H, p.W);
        const float hh = 1.f - t.lh, hw = 1.f - t.lw;
        const float mk = p.mask ? p.mask[m * (int64_t)(p.dg * taps) + g * taps + k] : 1.f;
        float s_dy = 0.f, s_dx = 0.f, s_mk = 0.f;
        for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
            const float gv = ld1(gc + c);
            const float v1 = t.ok1 ? ld1(img + ((int64_t)t.h_low * p.W + t.w_low) * p.C + c) : 0.f;
            const float v2 = t.ok2 ? ld1(img + ((int64_t)t.h_low * p.W + t.w_low + 1) * p.C + c) : 0.f;
            const float v3 = t.ok3 ? ld1(img + ((int64_t)(t.h_low + 1) * p.W + t.w_low) * p.C + c) : 0.f;
            const float v4 = t.ok4 ? ld1(img + ((int64_t)(t.h_low + 1) * p.W + t.w_low + 1) * p.C + c) : 0.f;
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

// Offset / mask gradients alone (no scatter): one wave per (pixel, tap), float4 per lane = 256 channels per pass,
// one wave reduction at the end.  Streams gcol once; the four corner rows of x come from L2.
template <typename T>
__global__ __launch_bounds__(256) void deform_goffset_kernel(DcnParams p, const T *__restrict__ gcol,
                                                             float *__restrict__ goffset, float *__restrict__ gmask)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int taps = p.kh * p.kw;
    if (task >= p.M * taps) return;
    int64_t m; int k, b, ho, wo;
    decode(p, task, m, k, b, ho, wo);
    const int ky = k / p.kw, kx = k % p.kw;
    const T *img = static_cast<const T *>(p.x) + (int64_t)b * p.H * p.W * p.C;
    const T *gc = gcol + (m * taps + k) * p.C;
    const float hs = (float)(ho * p.stride - p.pad + ky * p.dil) + p.offset[(m * taps + k) * 2];
    const float ws = (float)(wo * p.stride - p.pad + kx * p.dil) + p.offset[(m * taps + k) * 2 + 1];
    const Tap t = make_tap(hs, ws, p.H, p.W);
    const float hh = 1.f - t.lh, hw = 1.f - t.lw;
    const float mk = p.mask ? p.mask[m * taps + k] : 1.f;
    const T *r0 = img + ((int64_t)t.h_low * p.W + t.w_low) * p.C, *r1 = r0 + (int64_t)p.W * p.C;
    float s_dy = 0.f, s_dx = 0.f, s_mk = 0.f;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = lane * 4; c < p.C; c += 256) {
        const float4 gv = ld4v(gc + c);
        const float4 v1 = t.ok1 ? ld4v(r0 + c) : z;
        const float4 v2 = t.ok2 ? ld4v(r0 + p.C + c) : z;
        const float4 v3 = t.ok3 ? ld4v(r1 + c) : z;
        const float4 v4 = t.ok4 ? ld4v(r1 + p.C + c) : z;
        auto acc = [&](float g, float a1, float a2, float a3, float a4) {
            s_dy += g * mk * (hw * (a3 - a1) + t.lw * (a4 - a2));
            s_dx += g * mk * (hh * (a2 - a1) + t.lh * (a4 - a3));
            s_mk += g * (hh * hw * a1 + hh * t.lw * a2 + t.lh * hw * a3 + t.lh * t.lw * a4);
        };
        acc(gv.x, v1.x, v2.x, v3.x, v4.x);
        acc(gv.y, v1.y, v2.y, v3.y, v4.y);
        acc(gv.z, v1.z, v2.z, v3.z, v4.z);
        acc(gv.w, v1.w, v2.w, v3.w, v4.w);
    }
    s_dy = htd::wave_sum(s_dy);
    s_dx = htd::wave_sum(s_dx);
    if (lane == 0 && goffset) {
        goffset[(m * taps + k) * 2] = s_dy;
        goffset[(m * taps + k) * 2 + 1] = s_dx;
    }
    if (gmask) {
        s_mk = htd::wave_sum(s_mk);
        if (lane == 0) gmask[m * taps + k] = s_mk;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-owned col2im.  The direct kernel above issues four global float atomics per (pixel, tap, channel) and runs at
// the chip's float-atomic rate (1.3 TB/s of added bytes); LDS float atomics are no way out (ds_add_f32 retires one
// wave-instruction per ~165 cycles per CU).  Here a workgroup owns TILE x TILE output pixels and accumulates one
// 64-channel slice at a time in an LDS window with PLAIN read-modify-writes, made race-free by ownership: every
// window row belongs to one wave (row mod waves).  Once per workgroup the (pixel, tap) items are split into their top
// and bottom bilinear halves and bucketed by the window row they hit; in the slice loop each wave walks only its own
// bucket, so no two waves ever touch the same LDS word.  Halves that fall outside the window (offsets beyond the
// margin) are kept in a common bucket and go to global memory with atomics, so every offset is handled.  The window
// is flushed with one global atomic per touched (pixel, channel).  Offset / mask gradients come from a separate
// streaming pass (deform_goffset_kernel: one wave reduction per (pixel, tap) over all channels, no atomics).
struct HalfRec {          // one bilinear half (two corners of one row) of a (pixel, tap) item; 16 B
    int dst;              // owned: float index of the left corner in the LDS window; common: pixel index in the image
    int src;              // row of gcol: m * taps + k
    float wa, wb;         // mask * row weight * column weight of the left / right corner (0 when outside the image)
};

constexpr int RW_WAVES = 8;

template <typename T>
__global__ __launch_bounds__(RW_WAVES * 64) void deform_col2im_rows_kernel(
    DcnParams p, const T *__restrict__ gcol, float *__restrict__ gx, int tile, int tiles_x, int tiles_y, int WH,
    int WW, int margin, int slices_per_block)
{
    extern __shared__ float win[];                                  // [WH*WW][64] | HalfRec[2*items] | counters
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int taps = p.kh * p.kw;
    const int items = tile * tile * taps;
    HalfRec *recs = reinterpret_cast<HalfRec *>(win + WH * WW * 64);
    int *cnt = reinterpret_cast<int *>(recs + 2 * items);           // [RW_WAVES + 1] counts, then starts, then cursors
    int *start = cnt + RW_WAVES + 1, *cursor = start + RW_WAVES + 2;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int y_org = ty * tile * p.stride - p.pad - margin, x_org = tx * tile * p.stride - p.pad - margin;
    float *gimg = gx + (int64_t)b * p.H * p.W * p.C;
    const int slices = p.C >> 6;
    const int s_begin = blockIdx.y * slices_per_block, s_end = min(slices, s_begin + slices_per_block);

    // ---- once per workgroup: split the items into halves, bucket them by the wave that owns their window row
    if (threadIdx.x <= RW_WAVES) cnt[threadIdx.x] = cursor[threadIdx.x] = 0;
    __syncthreads();
    for (int phase = 0; phase < 2; ++phase) {
        for (int item = threadIdx.x; item < items; item += blockDim.x) {
            const int pix = item / taps, k = item - pix * taps;
            const int oy = ty * tile + pix / tile, ox = tx * tile + pix % tile;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            const int64_t m = ((int64_t)b * p.Ho + oy) * p.Wo + ox;
            const int ky = k / p.kw, kx = k - ky * p.kw;
            const float hs = (float)(oy * p.stride - p.pad + ky * p.dil) + p.offset[(m * taps + k) * 2];
            const float ws = (float)(ox * p.stride - p.pad + kx * p.dil) + p.offset[(m * taps + k) * 2 + 1];
            const Tap t = make_tap(hs, ws, p.H, p.W);
            const float mk = p.mask ? p.mask[m * taps + k] : 1.f;
            const int lx = t.w_low - x_org;
            const bool cols_in = lx >= 0 && lx + 1 < WW;
            for (int half = 0; half < 2; ++half) {
                const bool oka = half ? t.ok3 : t.ok1, okb = half ? t.ok4 : t.ok2;
                if (!oka && !okb) continue;
                const int ly = t.h_low + half - y_org;
                const bool owned = cols_in && ly >= 0 && ly < WH;
                const int owner = owned ? ly % RW_WAVES : RW_WAVES;
                if (phase == 0) {
                    atomicAdd(&cnt[owner], 1);
                } else {
                    const int pos = start[owner] + atomicAdd(&cursor[owner], 1);
                    const float wy = mk * (half ? t.lh : 1.f - t.lh);
                    HalfRec r;
                    r.dst = owned ? (ly * WW + lx) * 64 : (t.h_low + half) * p.W + t.w_low;
                    r.src = (int)(m * taps + k);
                    r.wa = oka ? wy * (1.f - t.lw) : 0.f;
                    r.wb = okb ? wy * t.lw : 0.f;
                    recs[pos] = r;
                }
            }
        }
        __syncthreads();
        if (phase == 0) {
            if (threadIdx.x == 0) {
                int acc = 0;
                for (int w = 0; w <= RW_WAVES; ++w) { start[w] = acc; acc += cnt[w]; }
                start[RW_WAVES + 1] = acc;
            }
            __syncthreads();
        }
    }
    const HalfRec *own = recs + start[wave];
    const int n_own = cnt[wave];
    const HalfRec *common = recs + start[RW_WAVES];
    const int n_common = cnt[RW_WAVES];

    constexpr int FLY = 4;
    struct Batch { HalfRec r[FLY]; float gv[FLY]; };
    for (int sl = s_begin; sl < s_end; ++sl) {
        const T *gc = gcol + sl * 64 + lane;
        for (int i = threadIdx.x; i < WH * WW * 16; i += blockDim.x)
            reinterpret_cast<float4 *>(win)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        // this wave's rows: plain LDS adds, the next batch's records and gradient columns fetched ahead
        auto fetch = [&](const HalfRec *list, int n, int e0, int step, Batch &bt) {
#pragma unroll
            for (int f = 0; f < FLY; ++f) {
                const int e = e0 + f * step;
                bt.r[f].wa = bt.r[f].wb = 0.f;
                bt.r[f].dst = bt.r[f].src = 0;
                bt.gv[f] = 0.f;
                if (e < n) {
                    bt.r[f] = list[e];
                    bt.gv[f] = ld1(gc + (int64_t)bt.r[f].src * p.C);
                }
            }
        };
        if (n_own > 0) {
            Batch cur, nxt;
            fetch(own, n_own, 0, 1, cur);
            for (int e0 = 0; e0 < n_own; e0 += FLY) {
                fetch(own, n_own, e0 + FLY, 1, nxt);
#pragma unroll
                for (int f = 0; f < FLY; ++f) {
                    if (e0 + f >= n_own) continue;                  // (a padding slot must not touch another wave's row)
                    float *q = win + cur.r[f].dst + lane;
                    q[0] += cur.gv[f] * cur.r[f].wa;
                    q[64] += cur.gv[f] * cur.r[f].wb;
                }
                cur = nxt;
            }
        }
        // halves outside the window: global atomics, shared round-robin between the waves
        for (int e0 = wave; e0 < n_common; e0 += FLY * RW_WAVES) {
            Batch bt;
            fetch(common, n_common, e0, RW_WAVES, bt);
#pragma unroll
            for (int f = 0; f < FLY; ++f) {
                if (e0 + f * RW_WAVES >= n_common) continue;
                float *q = gimg + (int64_t)bt.r[f].dst * p.C + sl * 64 + lane;
                if (bt.r[f].wa != 0.f) atomicAdd(q, bt.gv[f] * bt.r[f].wa);
                if (bt.r[f].wb != 0.f) atomicAdd(q + p.C, bt.gv[f] * bt.r[f].wb);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < WH * WW * 64; i += blockDim.x) {
            const float v = win[i];
            if (v != 0.f) {
                const int pix = i >> 6, y = y_org + pix / WW, x = x_org + pix % WW;
                // window cells outside the image only ever receive 0 * gradient; the bounds test keeps a NaN / Inf
                // gradient from turning that into a write outside gx
                if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                    atomicAdd(gimg + ((int64_t)y * p.W + x) * p.C + sl * 64 + (i & 63), v);
            }
        }
        __syncthreads();
    }
}

int fill(DcnParams &p, const void *x, const float *offset, const float *mask, int B, int H, int W, int C, int kh,
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

template <typename T>
static int launch_im2col(const void *x, const float *offset, const float *mask, T *columns, int B, int H, int W, int C,
                         int kh, int kw, int stride, int pad, int dil, int deform_groups, void *stream)
{
    DcnParams p{};
    const int st = fill(p, x, offset, mask, B, H, W, C, kh, kw, stride, pad, dil, deform_groups);
    if (st) return st;
    HTD_REQUIRE(columns, "deform_im2col: null columns");
    const int64_t tasks = p.M * kh * kw;
    const int64_t blocks = htd::ceil_div(tasks, 4);
    HTD_REQUIRE(blocks < (1ll << 31), "deform_im2col: too many tasks");
    hipLaunchKernelGGL(deform_im2col_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, columns);
    return htd::check_launch("deform_im2col");
}

template <typename T>
static int launch_col2im(const void *x, const float *offset, const float *mask, const T *gcolumns, float *gx,
                         float *goffset, float *gmask, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                         int dil, int deform_groups, void *stream)
{
    DcnParams p{};
    const int st = fill(p, x, offset, mask, B, H, W, C, kh, kw, stride, pad, dil, deform_groups);
    if (st) return st;
    HTD_REQUIRE(gcolumns, "deform_col2im: null gradient columns");
    const int64_t tasks = p.M * kh * kw;
    const int64_t blocks = htd::ceil_div(tasks, 4);
    HTD_REQUIRE(blocks < (1ll << 31), "deform_col2im: too many tasks");
    // row-owned LDS accumulation: needs gx, 64-channel slices inside one deformable group, a window within the LDS
    // budget of two workgroups per CU, and 32-bit row / pixel indices
    static const bool direct_only = getenv("HTD_DCN_DIRECT_COL2IM") != nullptr;
    const int tile = stride == 1 ? 8 : 4, margin = 1;
    const int WH = (tile - 1) * stride + (kh - 1) * dil + 2 + 2 * margin, WW = (tile - 1) * stride + (kw - 1) * dil + 2 + 2 * margin;
    const int items = tile * tile * kh * kw;
    const size_t lds = (size_t)WH * WW * 64 * sizeof(float) + (size_t)2 * items * sizeof(HalfRec) +
                       (3 * (RW_WAVES + 2)) * sizeof(int);
    if (!direct_only && gx && C % 64 == 0 && deform_groups == 1 && stride <= 2 && lds <= 80 * 1024 &&
        p.M * kh * kw < (1ll << 31) && (int64_t)H * W < (1ll << 31)) {
        const int tiles_x = (int)htd::ceil_div(p.Wo, tile), tiles_y = (int)htd::ceil_div(p.Ho, tile);
        const int64_t nt = (int64_t)tiles_x * tiles_y * B;
        HTD_REQUIRE(nt < (1ll << 31), "deform_col2im: too many tiles");
        const int slices = C / 64;
        int groups_y = (int)std::min<int64_t>(slices, std::max<int64_t>(1, htd::ceil_div(2048, nt)));   // >= 8 workgroups per CU
        const int spb = (int)htd::ceil_div(slices, groups_y);
        groups_y = (int)htd::ceil_div(slices, spb);
        if (goffset || gmask)           // offset / mask gradients: their own streaming pass (no atomics)
            hipLaunchKernelGGL(deform_goffset_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p,
                               gcolumns, goffset, gmask);
        static bool lds_opt_in = false;
        if (!lds_opt_in) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(deform_col2im_rows_kernel<T>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            lds_opt_in = true;
        }
        hipLaunchKernelGGL(deform_col2im_rows_kernel<T>, dim3((unsigned)nt, (unsigned)groups_y), dim3(RW_WAVES * 64), lds,
                           (hipStream_t)stream, p, gcolumns, gx, tile, tiles_x, tiles_y, WH, WW, margin, spb);
        return htd::check_launch("deform_col2im(rows)");
    }
    hipLaunchKernelGGL(deform_col2im_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, gcolumns, gx,
                       goffset, gmask);
    return htd::check_launch("deform_col2im");
}

extern "C" int htd_deform_im2col(const float *x, const float *offset, const float *mask, float *columns, int B, int H,
                                 int W, int C, int kh, int kw, int stride, int pad, int dil, int deform_groups,
                                 void *stream)
{
    return launch_im2col<float>(x, offset, mask, columns, B, H, W, C, kh, kw, stride, pad, dil, deform_groups, stream);
}

extern "C" int htd_deform_col2im(const float *x, const float *offset, const float *mask, const float *gcolumns,
                                 float *gx, float *goffset, float *gmask, int B, int H, int W, int C, int kh, int kw,
                                 int stride, int pad, int dil, int deform_groups, void *stream)
{
    return launch_col2im<float>(x, offset, mask, gcolumns, gx, goffset, gmask, B, H, W, C, kh, kw, stride, pad, dil,
                                deform_groups, stream);
}

extern "C" int htd_deform_im2col_bf16(const void *x, const float *offset, const float *mask, void *columns, int B, int H,
                                      int W, int C, int kh, int kw, int stride, int pad, int dil, int deform_groups,
                                      void *stream)
{
    return launch_im2col<bf16_t>(x, offset, mask, static_cast<bf16_t *>(columns), B, H, W, C, kh, kw, stride, pad, dil,
                                 deform_groups, stream);
}

extern "C" int htd_deform_col2im_bf16(const void *x, const float *offset, const float *mask, const void *gcolumns,
                                      float *gx, float *goffset, float *gmask, int B, int H, int W, int C, int kh, int kw,
                                      int stride, int pad, int dil, int deform_groups, void *stream)
{
    return launch_col2im<bf16_t>(x, offset, mask, static_cast<const bf16_t *>(gcolumns), gx, goffset, gmask, B, H, W, C, kh,
                                 kw, stride, pad, dil, deform_groups, stream);
}
