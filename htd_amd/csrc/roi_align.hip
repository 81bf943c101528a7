// RoIAlign forward / backward for NHWC fp32 feature maps on gfx950.
//
// One wavefront (64 lanes) owns one (RoI, output bin, 256-channel chunk).  The adaptive
// sampling grid of a bin is a product grid and bilinear weights factor per axis, so the
// bin's value is   sum_r sum_c Wy[r] * Wx[c] * feat[r][c][:]   with
//   Wy[r] = sum_iy valid(y_iy) * max(0, 1 - |clamp(y_iy, 0, H-1) - r|)
// i.e. every pixel of the bin's footprint is read exactly ONCE (the per-sample form of
// mmcv's kernel reads it up to 4*grid_h*grid_w/footprint times).  Lanes run along C, so
// every pixel access is one contiguous 1 KiB (fwd, float4/lane) or 256 B (bwd atomics,
// the full-rate shape for global float atomics) wave transaction.  HBM-bound by design.
#include <algorithm>

#include "common.h"

namespace {

struct RoiGeom {
    float start_h, start_w, bin_h, bin_w;
    int grid_h, grid_w, batch;
    float inv_count;
};

__device__ __forceinline__ RoiGeom roi_geometry(const float *roi, float scale, int ph, int pw,
                                                int sampling_ratio, int aligned)
{
    RoiGeom g;
    const float off = aligned ? 0.5f : 0.f;
    g.batch = (int)roi[0];
    g.start_w = roi[1] * scale - off;
    g.start_h = roi[2] * scale - off;
    const float end_w = roi[3] * scale - off, end_h = roi[4] * scale - off;
    float rw = end_w - g.start_w, rh = end_h - g.start_h;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    g.bin_h = rh / (float)ph;
    g.bin_w = rw / (float)pw;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
    const int cnt = g.grid_h * g.grid_w;
    g.inv_count = 1.f / (float)(cnt > 1 ? cnt : 1);
    return g;
}

// Weight that pixel index `p` receives from the `grid` samples of one bin along one axis.
__device__ __forceinline__ float axis_weight(float start, float bin, int bin_idx, int grid, int p,
                                             int extent)
{
    float w = 0.f;
    const float step = bin / (float)grid;
    const float base = start + (float)bin_idx * bin;
    for (int s = 0; s < grid; ++s) {
        const float v = base + ((float)s + .5f) * step;
        if (v < -1.0f || v > (float)extent) continue;
        const float vc = fminf(fmaxf(v, 0.f), (float)(extent - 1));
        w += fmaxf(0.f, 1.f - fabsf(vc - (float)p));
    }
    return w;
}

// Pixel span [lo, hi] touched by one bin along one axis (may be empty: lo > hi).
__device__ __forceinline__ void axis_span(float start, float bin, int bin_idx, int grid, int extent,
                                          int &lo, int &hi)
{
    if (grid <= 0) { lo = 0; hi = -1; return; }
    const float step = bin / (float)grid;
    const float base = start + (float)bin_idx * bin;
    const float v0 = base + .5f * step, v1 = base + ((float)(grid - 1) + .5f) * step;
    const float c0 = fminf(fmaxf(fminf(v0, v1), 0.f), (float)(extent - 1));
    const float c1 = fminf(fmaxf(fmaxf(v0, v1), 0.f), (float)(extent - 1));
    lo = (int)c0;
    hi = min((int)c1 + 1, extent - 1);
}

__device__ __forceinline__ float lane_bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

template <bool BACKWARD>
__global__ __launch_bounds__(256) void roi_align_kernel(const float *__restrict__ feat_or_gout,
                                                        const float *__restrict__ rois,
                                                        const int64_t *__restrict__ roi_level, int level,
                                                        float *__restrict__ out_or_gfeat, int64_t n,
                                                        int B, int C, int H, int W, int ph, int pw,
                                                        float scale, int sampling_ratio, int aligned,
                                                        int chunks, int64_t tasks)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (task >= tasks) return;  // whole wave exits together
    const int chunk = (int)(task % chunks);
    const int64_t t2 = task / chunks;
    const int bin = (int)(t2 % (ph * pw));
    const int64_t ri = t2 / (ph * pw);
    const int64_t roi_row = ri;
    if (roi_level && roi_level[ri] != (int64_t)level) return;  // wave-uniform
    const int bi = bin / pw, bj = bin % pw;
    const RoiGeom g = roi_geometry(rois + 5 * roi_row, scale, ph, pw, sampling_ratio, aligned);

    int r0, r1, c0, c1;
    axis_span(g.start_h, g.bin_h, bi, g.grid_h, H, r0, r1);
    axis_span(g.start_w, g.bin_w, bj, g.grid_w, W, c0, c1);
    if (g.batch < 0 || g.batch >= B) { r0 = 0; r1 = -1; }  // never index outside the feature map

    const size_t img_base = (size_t)g.batch * H * W * C;
    const size_t bin_off = ((size_t)roi_row * ph * pw + bin) * C;

    if (!BACKWARD) {
        const int ch = chunk * 256 + lane * 4;
        const bool act = ch < C;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int rb = r0; rb <= r1; rb += 64) {
            const float wy_l = (rb + lane <= r1) ? axis_weight(g.start_h, g.bin_h, bi, g.grid_h, rb + lane, H) : 0.f;
            for (int cb = c0; cb <= c1; cb += 64) {
                const float wx_l = (cb + lane <= c1) ? axis_weight(g.start_w, g.bin_w, bj, g.grid_w, cb + lane, W) : 0.f;
                const int rn = min(64, r1 - rb + 1), cn = min(64, c1 - cb + 1);
                for (int r = 0; r < rn; ++r) {
                    const float wy = lane_bcast(wy_l, r);
                    const float *row = feat_or_gout + img_base + ((size_t)(rb + r) * W + cb) * C + ch;
#pragma unroll 4
                    for (int c = 0; c < cn; ++c) {
                        const float w = wy * lane_bcast(wx_l, c);
                        if (act) {
                            const float4 v = *reinterpret_cast<const float4 *>(row + (size_t)c * C);
                            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
                        }
                    }
                }
            }
        }
        if (act) {
            acc.x *= g.inv_count; acc.y *= g.inv_count; acc.z *= g.inv_count; acc.w *= g.inv_count;
            *reinterpret_cast<float4 *>(out_or_gfeat + bin_off + ch) = acc;
        }
    } else {
        float go[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = chunk * 256 + lane + 64 * k;
            go[k] = ch < C ? feat_or_gout[bin_off + ch] * g.inv_count : 0.f;
        }
        for (int rb = r0; rb <= r1; rb += 64) {
            const float wy_l = (rb + lane <= r1) ? axis_weight(g.start_h, g.bin_h, bi, g.grid_h, rb + lane, H) : 0.f;
            for (int cb = c0; cb <= c1; cb += 64) {
                const float wx_l = (cb + lane <= c1) ? axis_weight(g.start_w, g.bin_w, bj, g.grid_w, cb + lane, W) : 0.f;
                const int rn = min(64, r1 - rb + 1), cn = min(64, c1 - cb + 1);
                for (int r = 0; r < rn; ++r) {
                    const float wy = lane_bcast(wy_l, r);
                    if (wy == 0.f) continue;
                    float *row = out_or_gfeat + img_base + ((size_t)(rb + r) * W + cb) * C + chunk * 256 + lane;
                    for (int c = 0; c < cn; ++c) {
                        const float w = wy * lane_bcast(wx_l, c);
                        if (w == 0.f) continue;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (chunk * 256 + lane + 64 * k < C)
                                atomicAdd(row + (size_t)c * C + 64 * k, w * go[k]);
                    }
                }
            }
        }
    }
}

// Forward for a FEW RoIs (BA pools two dozen large ones from every level: a 300-pixel box on the stride-4 level is a
// 12 x 12-pixel footprint per bin, the whole image 48 x 29): one WORKGROUP per (RoI, bin, 256-channel chunk), its eight
// wavefronts take every eighth footprint row and the partial sums meet in LDS in a fixed order.  With one wavefront per bin
// such a launch is 1 176 long serial walks on a 256-CU chip.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void roi_align_fwd_split_kernel(const float *__restrict__ feat, const float *__restrict__ rois,
                                                                  const int64_t *__restrict__ roi_level, int level,
                                                                  float *__restrict__ out, int B, int C, int H, int W, int ph,
                                                                  int pw, float scale, int sampling_ratio, int aligned,
                                                                  int chunks)
{
    __shared__ float4 part[WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t task = blockIdx.x;
    const int chunk = (int)(task % chunks);
    const int64_t t2 = task / chunks;
    const int bin = (int)(t2 % (ph * pw));
    const int64_t ri = t2 / (ph * pw);
    if (roi_level && roi_level[ri] != (int64_t)level) return;  // block-uniform
    const int bi = bin / pw, bj = bin % pw;
    const RoiGeom g = roi_geometry(rois + 5 * ri, scale, ph, pw, sampling_ratio, aligned);
    int r0, r1, c0, c1;
    axis_span(g.start_h, g.bin_h, bi, g.grid_h, H, r0, r1);
    axis_span(g.start_w, g.bin_w, bj, g.grid_w, W, c0, c1);
    if (g.batch < 0 || g.batch >= B) { r0 = 0; r1 = -1; }
    const size_t img_base = (size_t)g.batch * H * W * C;
    const int ch = chunk * 256 + lane * 4;
    const bool act = ch < C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rb = r0; rb <= r1; rb += 64) {
        const float wy_l = (rb + lane <= r1) ? axis_weight(g.start_h, g.bin_h, bi, g.grid_h, rb + lane, H) : 0.f;
        const int rn = min(64, r1 - rb + 1);
        for (int cb = c0; cb <= c1; cb += 64) {
            const float wx_l = (cb + lane <= c1) ? axis_weight(g.start_w, g.bin_w, bj, g.grid_w, cb + lane, W) : 0.f;
            const int cn = min(64, c1 - cb + 1);
            for (int r = wave; r < rn; r += WAVES) {
                const float wy = lane_bcast(wy_l, r);
                const float *row = feat + img_base + ((size_t)(rb + r) * W + cb) * C + ch;
#pragma unroll 4
                for (int c = 0; c < cn; ++c) {
                    const float w = wy * lane_bcast(wx_l, c);
                    if (act) {
                        const float4 v = *reinterpret_cast<const float4 *>(row + (size_t)c * C);
                        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
                    }
                }
            }
        }
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && act) {
        float4 t = part[0][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            const float4 v = part[w][lane];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        t.x *= g.inv_count; t.y *= g.inv_count; t.z *= g.inv_count; t.w *= g.inv_count;
        *reinterpret_cast<float4 *>(out + ((size_t)ri * ph * pw + bin) * C + ch) = t;
    }
}

// Forward over ALL pyramid levels of SingleRoIExtractor in one launch: a wavefront reads its RoI's level and takes that
// level's map, size and scale from the table (four launches that each skip the other levels' RoIs start ~4x the waves).
struct LevelTable {
    const float *feat[8];
    int H[8], W[8];
    float scale[8];
};

__global__ __launch_bounds__(256) void roi_align_levels_fwd_kernel(LevelTable tab, const float *__restrict__ rois,
                                                                   const int64_t *__restrict__ roi_level, float *__restrict__ out,
                                                                   int64_t n, int B, int C, int L, int ph, int pw,
                                                                   int sampling_ratio, int aligned, int chunks, int64_t tasks,
                                                                   float *__restrict__ amax_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (task >= tasks) return;
    // amax_out: what the scalar holds now (read early; a wavefront that cannot raise it stays away from the atomic)
    const unsigned seen = amax_out ? *reinterpret_cast<const volatile unsigned *>(amax_out) : 0u;
    const int chunk = (int)(task % chunks);
    const int64_t t2 = task / chunks;
    const int bin = (int)(t2 % (ph * pw));
    const int64_t ri = t2 / (ph * pw);
    const int lv = (int)roi_level[ri];
    const int ch = chunk * 256 + lane * 4;
    const bool act = ch < C;
    const size_t bin_off = ((size_t)ri * ph * pw + bin) * C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    RoiGeom g{};
    g.inv_count = 0.f;
    if (lv >= 0 && lv < L) {                              // wave-uniform
        const int H = tab.H[lv], W = tab.W[lv];
        const float *feat = tab.feat[lv];
        const int bi = bin / pw, bj = bin % pw;
        g = roi_geometry(rois + 5 * ri, tab.scale[lv], ph, pw, sampling_ratio, aligned);
        int r0, r1, c0, c1;
        axis_span(g.start_h, g.bin_h, bi, g.grid_h, H, r0, r1);
        axis_span(g.start_w, g.bin_w, bj, g.grid_w, W, c0, c1);
        if (g.batch < 0 || g.batch >= B) { r0 = 0; r1 = -1; }
        const size_t img_base = (size_t)g.batch * H * W * C;
        for (int rb = r0; rb <= r1; rb += 64) {
            const float wy_l = (rb + lane <= r1) ? axis_weight(g.start_h, g.bin_h, bi, g.grid_h, rb + lane, H) : 0.f;
            for (int cb = c0; cb <= c1; cb += 64) {
                const float wx_l = (cb + lane <= c1) ? axis_weight(g.start_w, g.bin_w, bj, g.grid_w, cb + lane, W) : 0.f;
                const int rn = min(64, r1 - rb + 1), cn = min(64, c1 - cb + 1);
                for (int r = 0; r < rn; ++r) {
                    const float wy = lane_bcast(wy_l, r);
                    const float *row = feat + img_base + ((size_t)(rb + r) * W + cb) * C + ch;
#pragma unroll 4
                    for (int c = 0; c < cn; ++c) {
                        const float w = wy * lane_bcast(wx_l, c);
                        if (act) {
                            const float4 v = *reinterpret_cast<const float4 *>(row + (size_t)c * C);
                            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
                        }
                    }
                }
            }
        }
    }
    if (act) {
        acc.x *= g.inv_count; acc.y *= g.inv_count; acc.z *= g.inv_count; acc.w *= g.inv_count;
        *reinterpret_cast<float4 *>(out + bin_off + ch) = acc;
    }
    if (amax_out) {                                       // wave-uniform: max |out| of the launch (NaN sorts above everything)
        const bool nan = act && ((acc.x != acc.x) | (acc.y != acc.y) | (acc.z != acc.z) | (acc.w != acc.w));
        const float m = act ? fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w))) : 0.f;
        unsigned bits = nan ? 0x7fc00000u : __float_as_uint(m);
        for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o));
        if (lane == 0 && bits > seen)
            __hip_atomic_fetch_max(reinterpret_cast<unsigned *>(amax_out), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Forward of EVERY RoI on EVERY level in one launch (AdptRoIExtractor / BA, adaptative_roi_extractor.py:66-76: four RoIAlign
// calls over the same RoI list): task = (level, RoI, bin, chunk), level-major so that the wavefronts of a workgroup read one
// map; out[l] is level l's (n, ph, pw, C) tensor.  WAVES = 1: a wavefront per task; WAVES = 8: a workgroup per task whose
// wavefronts take every eighth footprint row (few RoIs with large footprints: the split form above).
struct AllLevelsOut {
    float *out[8];
};

template <int WAVES>
__global__ __launch_bounds__(WAVES == 1 ? 256 : 64 * WAVES) void roi_align_all_levels_fwd_kernel(
    LevelTable tab, AllLevelsOut outs, const float *__restrict__ rois, int64_t n, int B, int C, int L, int ph, int pw,
    int sampling_ratio, int aligned, int chunks, int64_t tasks)
{
    __shared__ float4 part[WAVES == 1 ? 1 : WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t task = WAVES == 1 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
    if (WAVES == 1 && task >= tasks) return;
    const int chunk = (int)(task % chunks);
    int64_t t2 = task / chunks;
    const int bin = (int)(t2 % (ph * pw));
    t2 /= ph * pw;
    const int64_t ri = t2 % n;
    const int lv = (int)(t2 / n);
    const int H = tab.H[lv], W = tab.W[lv];
    const float *feat = tab.feat[lv];
    const int bi = bin / pw, bj = bin % pw;
    const RoiGeom g = roi_geometry(rois + 5 * ri, tab.scale[lv], ph, pw, sampling_ratio, aligned);
    int r0, r1, c0, c1;
    axis_span(g.start_h, g.bin_h, bi, g.grid_h, H, r0, r1);
    axis_span(g.start_w, g.bin_w, bj, g.grid_w, W, c0, c1);
    if (g.batch < 0 || g.batch >= B) { r0 = 0; r1 = -1; }
    const size_t img_base = (size_t)g.batch * H * W * C;
    const int ch = chunk * 256 + lane * 4;
    const bool act = ch < C;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rb = r0; rb <= r1; rb += 64) {
        const float wy_l = (rb + lane <= r1) ? axis_weight(g.start_h, g.bin_h, bi, g.grid_h, rb + lane, H) : 0.f;
        const int rn = min(64, r1 - rb + 1);
        for (int cb = c0; cb <= c1; cb += 64) {
            const float wx_l = (cb + lane <= c1) ? axis_weight(g.start_w, g.bin_w, bj, g.grid_w, cb + lane, W) : 0.f;
            const int cn = min(64, c1 - cb + 1);
            for (int r = (WAVES == 1 ? 0 : wave); r < rn; r += WAVES) {
                const float wy = lane_bcast(wy_l, r);
                const float *row = feat + img_base + ((size_t)(rb + r) * W + cb) * C + ch;
#pragma unroll 4
                for (int c = 0; c < cn; ++c) {
                    const float w = wy * lane_bcast(wx_l, c);
                    if (act) {
                        const float4 v = *reinterpret_cast<const float4 *>(row + (size_t)c * C);
                        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
                    }
                }
            }
        }
    }
    float *dst = outs.out[lv] + ((size_t)ri * ph * pw + bin) * C + ch;
    if constexpr (WAVES == 1) {
        if (act) {
            acc.x *= g.inv_count; acc.y *= g.inv_count; acc.z *= g.inv_count; acc.w *= g.inv_count;
            *reinterpret_cast<float4 *>(dst) = acc;
        }
    } else {
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && act) {
            float4 t = part[0][lane];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                const float4 v = part[w][lane];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            t.x *= g.inv_count; t.y *= g.inv_count; t.z *= g.inv_count; t.w *= g.inv_count;
            *reinterpret_cast<float4 *>(dst) = t;
        }
    }
}

// Backward, row-wise: one wavefront owns (RoI, footprint row mod row_slots, 256-channel chunk); row_slots grows when
// there are few RoIs (BA pools two dozen large RoIs from every level) so that the launch still fills the chip.  For its row r it
// first folds the bins along y,  T[q][:] = sum_p Wy[p][r] * gout[p][q][:] / count  (registers), then walks the row's
// columns with  gfeat[r][c][:] += sum_q Wx[q][c] * T[q][:]:  ONE atomic per footprint pixel and channel.  The bin-wise
// form above issues one per (bin, pixel of the bin's span), i.e. about twice as many, because neighbouring bins'
// bilinear spans overlap -- and the 1.3 TB/s float-atomic rate is what bounds this kernel.
constexpr int MAXP = 8;

__global__ __launch_bounds__(256) void roi_align_bwd_rows_kernel(const float *__restrict__ gout,
                                                                 const float *__restrict__ rois,
                                                                 const int64_t *__restrict__ roi_level, int level,
                                                                 float *__restrict__ gfeat, int64_t n, int B, int C,
                                                                 int H, int W, int ph, int pw, float scale,
                                                                 int sampling_ratio, int aligned, int chunks,
                                                                 int row_slots, int64_t tasks)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (task >= tasks) return;  // whole wave exits together
    const int chunk = (int)(task % chunks);
    const int64_t t2 = task / chunks;
    const int slot = (int)(t2 % row_slots);
    const int64_t ri = t2 / row_slots;
    if (roi_level && roi_level[ri] != (int64_t)level) return;  // wave-uniform
    const RoiGeom g = roi_geometry(rois + 5 * ri, scale, ph, pw, sampling_ratio, aligned);
    if (g.batch < 0 || g.batch >= B) return;
    int r_lo, r_hi, c_lo, c_hi, t0, t1;
    axis_span(g.start_h, g.bin_h, 0, g.grid_h, H, r_lo, t0);
    axis_span(g.start_h, g.bin_h, ph - 1, g.grid_h, H, t1, r_hi);
    if (t0 < r_lo || r_hi < t1) return;                          // empty spans (degenerate RoI)
    axis_span(g.start_w, g.bin_w, 0, g.grid_w, W, c_lo, t0);
    axis_span(g.start_w, g.bin_w, pw - 1, g.grid_w, W, t1, c_hi);
    if (t0 < c_lo || c_hi < t1) return;
    const size_t img_base = (size_t)g.batch * H * W * C;
    const float *go = gout + (size_t)ri * ph * pw * C + chunk * 256 + lane;
    bool chk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) chk[k] = chunk * 256 + lane + 64 * k < C;

    for (int r = r_lo + slot; r <= r_hi; r += row_slots) {
        const float wy_l = lane < ph ? axis_weight(g.start_h, g.bin_h, lane, g.grid_h, r, H) : 0.f;
        float T[MAXP][4];
#pragma unroll
        for (int q = 0; q < MAXP; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) T[q][k] = 0.f;
        bool any_row = false;
        for (int p = 0; p < ph; ++p) {
            const float wy = lane_bcast(wy_l, p) * g.inv_count;
            if (wy == 0.f) continue;
            any_row = true;
#pragma unroll
            for (int q = 0; q < MAXP; ++q)
                if (q < pw) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (chk[k]) T[q][k] += wy * go[((size_t)p * pw + q) * C + 64 * k];
                }
        }
        if (!any_row) continue;
        for (int cb = c_lo; cb <= c_hi; cb += 64) {
            float wx_l[MAXP];
#pragma unroll
            for (int q = 0; q < MAXP; ++q)
                wx_l[q] = (q < pw && cb + lane <= c_hi) ? axis_weight(g.start_w, g.bin_w, q, g.grid_w, cb + lane, W) : 0.f;
            const int cn = min(64, c_hi - cb + 1);
            float *row = gfeat + img_base + ((size_t)r * W + cb) * C + chunk * 256 + lane;
            for (int c = 0; c < cn; ++c) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                bool any = false;
#pragma unroll
                for (int q = 0; q < MAXP; ++q) {
                    const float w = lane_bcast(wx_l[q], c);
                    if (w != 0.f) {
                        any = true;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += w * T[q][k];
                    }
                }
                if (!any) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (chk[k]) atomicAdd(row + (size_t)c * C + 64 * k, v[k]);
            }
        }
    }
}

// ---- backward, gather form: no atomics, bit-stable ---------------------------------------------------------------
// The scatter kernels above add every RoI's contribution with float atomics: ~1.3 TB/s of added bytes is the chip's
// ceiling for that, and the order of the additions -- hence the last bits of the gradient -- changes from run to run.
// Here a wavefront OWNS a strip of GW_TILE consecutive pixels of one feature-map row (all channels of a 64*CPL-wide
// chunk), walks the RoIs whose footprint covers the strip in ascending RoI order, and writes every pixel once:
//   gfeat[y][x][:] (+)= sum_r sum_q Wx_r[q][x] * ( sum_p Wy_r[p][y] * gout[r][p][q][:] / count_r )
// Footprint boxes are precomputed per RoI (roi_bbox_kernel: 16 bytes each); a strip finds its RoIs with one ballot per
// 64 boxes.  Uncovered strips are written as zeros (accumulate = 0: the map needs no memset) or left alone
// (accumulate = 1: the map already holds another consumer's gradient).
constexpr int GW_TILE = 8;

struct RoiBox { short b, r_lo, r_hi, c_lo, c_hi, pad0, pad1, pad2; };      // b < 0: not on this level / degenerate

__global__ __launch_bounds__(256) void roi_bbox_kernel(const float *__restrict__ rois, const int64_t *__restrict__ roi_level,
                                                       int level, RoiBox *__restrict__ box, int64_t n, int B, int H, int W,
                                                       int ph, int pw, float scale, int sampling_ratio, int aligned)
{
    const int64_t ri = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ri >= n) return;
    RoiBox o{};
    o.b = -1;
    if (!roi_level || roi_level[ri] == (int64_t)level) {
        const RoiGeom g = roi_geometry(rois + 5 * ri, scale, ph, pw, sampling_ratio, aligned);
        int r_lo, r_hi, c_lo, c_hi, t0, t1;
        axis_span(g.start_h, g.bin_h, 0, g.grid_h, H, r_lo, t0);
        axis_span(g.start_h, g.bin_h, ph - 1, g.grid_h, H, t1, r_hi);
        bool ok = g.batch >= 0 && g.batch < B && !(t0 < r_lo || r_hi < t1);
        axis_span(g.start_w, g.bin_w, 0, g.grid_w, W, c_lo, t0);
        axis_span(g.start_w, g.bin_w, pw - 1, g.grid_w, W, t1, c_hi);
        ok = ok && !(t0 < c_lo || c_hi < t1);
        if (ok) { o.b = (short)g.batch; o.r_lo = (short)r_lo; o.r_hi = (short)r_hi; o.c_lo = (short)c_lo; o.c_hi = (short)c_hi; }
    }
    box[ri] = o;
}

// CPL channels per lane (64 * CPL-channel chunks), ROWS feature-map rows per wavefront: a tile is ROWS x GW_TILE pixels.
// Taller tiles share the gout loads of a bin row between the feature rows it reaches (a 4-row tile reads ~3.3x fewer
// bytes per RoI than four 1-row strips: the kernel is bound by those L2 reads), at ROWS * GW_TILE * CPL accumulators.
// RS > 1 (round 4): the RS wavefronts of a group share ONE tile and deal its RoIs among themselves -- wavefront `part` takes the
// hits with RoI index = part (mod RS) -- and their partial sums meet in LDS in part order (fixed: bit-stable like RS = 1, other
// last bits).  What bounds this kernel is the CHAIN of RoIs a tile walks one after the other (geometry -> weights -> gout rows
// -> sums: a few microseconds of dependent latency each): BA pools every positive RoI from EVERY level, so a strip of the
// coarse maps walks all 128 RoIs of its image; RS wavefronts walk a quarter each.  `active` = the task exists (every
// wavefront of the block reaches the barrier).
// FOLD (round 4): the bins of a RoI are folded along y ONCE per (RoI, map row) by roi_fold_kernel into tbuf [n][H][pw][C] and a
// strip only fetches the one or two folded vectors its eight pixels touch -- a strip used to fetch and fold the 7-14 bin
// vectors itself, the ~20 strips of a large RoI's row each doing the same work (2 KB instead of 7-14 KB of L2 traffic and
// none of the fold's arithmetic per (strip, RoI) pair).  Same sums in the same order: bit-identical to FOLD = false.
template <int CPL, int ROWS, int RS, bool FOLD>
__device__ __forceinline__ void gather_tile(const float *__restrict__ gout, const float *__restrict__ rois,
                                            const RoiBox *__restrict__ box, float *__restrict__ gfeat, int64_t n, int B, int C,
                                            int H, int W, int ph, int pw, float scale, int sampling_ratio, int aligned,
                                            int accumulate, int chunks, int segs, int hts, int64_t task, bool active,
                                            const float *__restrict__ tbuf)
{
    static_assert(!FOLD || ROWS == 1, "folded rows: one-row tiles");
    static_assert(ROWS * MAXP <= 64, "one lane per (row, bin) weight");
    static_assert(RS == 1 || ROWS == 1, "the RoI split keeps one-row tiles");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, part = wave % RS;
    // the RoIs of this wavefront: bits i of a 64-box ballot with i = part (mod RS) (the box index base + i has base % 64 == 0)
    constexpr unsigned long long kEvery = RS == 4 ? 0x1111111111111111ull : (RS == 2 ? 0x5555555555555555ull : ~0ull);
    const unsigned long long mine = kEvery << part;
    const int chunk = (int)(task % chunks);
    int64_t t2 = task / chunks;
    const int seg = (int)(t2 % segs);
    t2 /= segs;
    const int ht = (int)(t2 % hts), b = (int)(t2 / hts);
    const int y0 = ht * ROWS, y1 = min(H, y0 + ROWS) - 1;
    const int x0 = seg * GW_TILE, x1 = min(W, x0 + GW_TILE) - 1;
    const int ch = chunk * 64 * CPL + lane * CPL;
    const bool act = ch < C;

    float acc[ROWS][GW_TILE][CPL];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int i = 0; i < GW_TILE; ++i)
#pragma unroll
            for (int k = 0; k < CPL; ++k) acc[r][i][k] = 0.f;
    bool touched = false;

    for (int64_t base = 0; base < (active ? n : 0); base += 64) {
        bool hit = false;
        if (base + lane < n) {
            const RoiBox o = box[base + lane];
            hit = o.b == b && o.r_lo <= y1 && y0 <= o.r_hi && o.c_lo <= x1 && o.c_hi >= x0;
        }
        unsigned long long mask = __ballot(hit) & mine;
        while (mask) {
            const int src = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int64_t ri = base + src;              // wave-uniform, ascending: the summation order is fixed
            touched = true;
            const RoiGeom g = roi_geometry(rois + 5 * ri, scale, ph, pw, sampling_ratio, aligned);
            if constexpr (FOLD) {
                static_assert(GW_TILE * MAXP == 64, "one lane per (bin, pixel) weight");
                const int xi_f = lane % GW_TILE, q_f = lane / GW_TILE;
                const float wx_f = (x0 + xi_f <= x1 && q_f < pw) ? axis_weight(g.start_w, g.bin_w, q_f, g.grid_w, x0 + xi_f, W) : 0.f;
                const float *tp = tbuf + (((size_t)ri * H + y0) * pw) * C + ch;
#pragma unroll
                for (int q = 0; q < MAXP; ++q) {
                    if (q >= pw) continue;
                    if (__ballot(q_f == q && wx_f != 0.f) == 0ull) continue;       // no pixel of the strip takes bin q (wave-uniform)
                    float v[CPL];
#pragma unroll
                    for (int k = 0; k < CPL; ++k) v[k] = 0.f;
                    if (act) {
                        if constexpr (CPL == 4) {
                            const float4 t4 = *reinterpret_cast<const float4 *>(tp + (size_t)q * C);
                            v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
                        } else {
#pragma unroll
                            for (int k = 0; k < CPL; ++k) v[k] = tp[(size_t)q * C + k];
                        }
                    }
#pragma unroll
                    for (int xi = 0; xi < GW_TILE; ++xi) {
                        const float w = lane_bcast(wx_f, q * GW_TILE + xi);
                        if (w != 0.f) {
#pragma unroll
                            for (int k = 0; k < CPL; ++k) acc[0][xi][k] += w * v[k];
                        }
                    }
                }
                continue;
            }
            // y weights: lane (r, p) = (lane / MAXP, lane % MAXP) holds Wy[p][y0 + r] / count
            const int r_l = lane / MAXP, p_l = lane % MAXP;
            const float wy_l = (r_l < ROWS && p_l < ph && y0 + r_l <= y1)
                                   ? axis_weight(g.start_h, g.bin_h, p_l, g.grid_h, y0 + r_l, H) * g.inv_count : 0.f;
            // fold the bins along y for every row of the tile:  T[r][q] = sum_p Wy[p][y0 + r] * gout[p][q]
            float T[ROWS][MAXP][CPL];
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
#pragma unroll
                for (int q = 0; q < MAXP; ++q)
#pragma unroll
                    for (int k = 0; k < CPL; ++k) T[r][q][k] = 0.f;
            const float *go = gout + (size_t)ri * ph * pw * C + ch;
            for (int p = 0; p < ph; ++p) {
                float wy[ROWS];
                bool any = false;
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    wy[r] = lane_bcast(wy_l, r * MAXP + p);
                    any = any || wy[r] != 0.f;
                }
                if (!any) continue;
#pragma unroll
                for (int q = 0; q < MAXP; ++q)
                    if (q < pw && act) {
                        float v[CPL];
                        if constexpr (CPL == 4) {
                            const float4 t = *reinterpret_cast<const float4 *>(go + ((size_t)p * pw + q) * C);
                            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                        } else if constexpr (CPL == 2) {
                            const float2 t = *reinterpret_cast<const float2 *>(go + ((size_t)p * pw + q) * C);
                            v[0] = t.x; v[1] = t.y;
                        } else {
                            v[0] = go[((size_t)p * pw + q) * C];
                        }
#pragma unroll
                        for (int r = 0; r < ROWS; ++r)
#pragma unroll
                            for (int k = 0; k < CPL; ++k) T[r][q][k] += wy[r] * v[k];
                    }
            }
            // x weights of the tile: lane (q, xi) = (lane / GW_TILE, lane % GW_TILE) holds Wx[q][x0 + xi]
            static_assert(GW_TILE * MAXP == 64, "one lane per (bin, pixel) weight");
            const int xi_l = lane % GW_TILE, q_l = lane / GW_TILE;
            const float wx_l = (x0 + xi_l <= x1 && q_l < pw) ? axis_weight(g.start_w, g.bin_w, q_l, g.grid_w, x0 + xi_l, W) : 0.f;
#pragma unroll
            for (int xi = 0; xi < GW_TILE; ++xi) {
#pragma unroll
                for (int q = 0; q < MAXP; ++q) {
                    if (q >= pw) continue;
                    const float w = lane_bcast(wx_l, q * GW_TILE + xi);
                    if (w != 0.f) {
#pragma unroll
                        for (int r = 0; r < ROWS; ++r)
#pragma unroll
                            for (int k = 0; k < CPL; ++k) acc[r][xi][k] += w * T[r][q][k];
                    }
                }
            }
        }
    }
    if constexpr (RS > 1) {
        // partial sums of parts 1 .. RS - 1 -> LDS [wave][element][lane], added by part 0 in part order
        __shared__ float lsum[4][GW_TILE * CPL][64];
        __shared__ int ltouched[4];
        if (part > 0) {
#pragma unroll
            for (int i = 0; i < GW_TILE; ++i)
#pragma unroll
                for (int k = 0; k < CPL; ++k) lsum[wave][i * CPL + k][lane] = acc[0][i][k];
            if (lane == 0) ltouched[wave] = touched ? 1 : 0;
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int w = 1; w < RS; ++w) {
#pragma unroll
            for (int i = 0; i < GW_TILE; ++i)
#pragma unroll
                for (int k = 0; k < CPL; ++k) acc[0][i][k] += lsum[wave + w][i * CPL + k][lane];
            touched = touched || ltouched[wave + w] != 0;
        }
    }
    if (!active || !act || (accumulate && !touched)) return;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (y0 + r > y1) continue;
        float *row = gfeat + (((size_t)b * H + y0 + r) * W + x0) * C + ch;
#pragma unroll
        for (int xi = 0; xi < GW_TILE; ++xi) {
            if (x0 + xi > x1) continue;
            float *dst = row + (size_t)xi * C;
#pragma unroll
            for (int k = 0; k < CPL; ++k) dst[k] = accumulate ? dst[k] + acc[r][xi][k] : acc[r][xi][k];
        }
    }
}

template <int CPL, int ROWS, int RS>
__global__ __launch_bounds__(256, (RS > 1 ? 4 : 1)) void roi_align_bwd_gather_kernel(const float *__restrict__ gout, const float *__restrict__ rois,
                                                                   const RoiBox *__restrict__ box, float *__restrict__ gfeat,
                                                                   int64_t n, int B, int C, int H, int W, int ph, int pw,
                                                                   float scale, int sampling_ratio, int aligned, int accumulate,
                                                                   int chunks, int segs, int hts, int64_t tasks)
{
    const int64_t task = (int64_t)blockIdx.x * (4 / RS) + (threadIdx.x >> 6) / RS;
    if (RS == 1 && task >= tasks) return;               // whole wave exits together
    gather_tile<CPL, ROWS, RS, false>(gout, rois, box, gfeat, n, B, C, H, W, ph, pw, scale, sampling_ratio, aligned, accumulate,
                                      chunks, segs, hts, task < tasks ? task : 0, task < tasks, nullptr);
}

// All pyramid levels of one extractor in ONE launch.  Launched level by level, the coarse maps set the time: P5 has 600
// strips for 5 120 wavefront slots, and each of them walks ~50 large RoIs one after the other (a few microseconds of
// dependent latency per RoI) while the rest of the chip idles; P2 then runs alone, 33 600 short strips.  With the levels in
// one grid -- coarsest first, so the long chains start at once -- the fine levels' strips fill the idle slots.
constexpr int GL_MAX = 6;
struct GatherLevels {
    float *gfeat[GL_MAX];
    const float *gout[GL_MAX];       // per slot (BA: every level has its own pooled tensor); NULL: the launch's shared grad_out
    float *tbuf[GL_MAX];             // FOLD: the slot's folded bins [n][H][pw][C] (roi_fold_kernel)
    int64_t ftask0[GL_MAX + 1];      // roi_fold_kernel: first task of slot k (n * H[k] * chunks tasks each)
    const RoiBox *box[GL_MAX];
    int H[GL_MAX], W[GL_MAX], segs[GL_MAX], hts[GL_MAX], accumulate[GL_MAX], level[GL_MAX];
    float scale[GL_MAX];
    int64_t task0[GL_MAX + 1];       // first task of slot k (slots in launch order: coarsest level first)
    int slots;
};

// T[ri][y][q][:] = sum_p Wy_ri[p][y] * gout_ri[p][q][:] / count_ri for every map row y of RoI ri's footprint, every level slot of
// the table: one wavefront per (slot, RoI, row, 256-channel chunk); rows outside the footprint exit at once.
__global__ __launch_bounds__(256) void roi_fold_kernel(const float *__restrict__ gout, const float *__restrict__ rois, GatherLevels t,
                                                       int64_t n, int C, int ph, int pw, int sampling_ratio, int aligned, int chunks)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= t.ftask0[t.slots]) return;
    int k = 0;
#pragma unroll
    for (int i = 1; i < GL_MAX; ++i)
        if (i < t.slots && task >= t.ftask0[i]) k = i;
    int64_t r = task - t.ftask0[k];
    const int chunk = (int)(r % chunks);
    r /= chunks;
    const int H = t.H[k];
    const int y = (int)(r % H);
    const int64_t ri = r / H;
    const RoiBox o = t.box[k][ri];
    if (o.b < 0 || y < o.r_lo || y > o.r_hi) return;                 // wave-uniform
    const RoiGeom g = roi_geometry(rois + 5 * ri, t.scale[k], ph, pw, sampling_ratio, aligned);
    const float wy_l = lane < ph ? axis_weight(g.start_h, g.bin_h, lane, g.grid_h, y, H) * g.inv_count : 0.f;
    const int ch = chunk * 256 + lane * 4;
    if (ch >= C) return;
    float T[MAXP][4];
#pragma unroll
    for (int q = 0; q < MAXP; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) T[q][c] = 0.f;
    const float *go = (t.gout[k] ? t.gout[k] : gout) + (size_t)ri * ph * pw * C + ch;
    for (int p = 0; p < ph; ++p) {
        const float wy = lane_bcast(wy_l, p);
        if (wy == 0.f) continue;
#pragma unroll
        for (int q = 0; q < MAXP; ++q)
            if (q < pw) {
                const float4 v = *reinterpret_cast<const float4 *>(go + ((size_t)p * pw + q) * C);
                T[q][0] += wy * v.x; T[q][1] += wy * v.y; T[q][2] += wy * v.z; T[q][3] += wy * v.w;
            }
    }
    float *dst = t.tbuf[k] + (((size_t)ri * H + y) * pw) * C + ch;
#pragma unroll
    for (int q = 0; q < MAXP; ++q)
        if (q < pw) *reinterpret_cast<float4 *>(dst + (size_t)q * C) = make_float4(T[q][0], T[q][1], T[q][2], T[q][3]);
}

template <int CPL, int ROWS, int RS, bool FOLD>
__global__ __launch_bounds__(256, (RS > 1 ? 4 : 1)) void roi_align_bwd_gather_levels_kernel(const float *__restrict__ gout,
                                                                          const float *__restrict__ rois, GatherLevels t, int64_t n,
                                                                          int B, int C, int ph, int pw, int sampling_ratio,
                                                                          int aligned, int chunks)
{
    int64_t task = (int64_t)blockIdx.x * (4 / RS) + (threadIdx.x >> 6) / RS;
    const bool active = task < t.task0[t.slots];
    if (RS == 1 && !active) return;
    if (!active) task = 0;
    int k = 0;
#pragma unroll
    for (int i = 1; i < GL_MAX; ++i)
        if (i < t.slots && task >= t.task0[i]) k = i;
    gather_tile<CPL, ROWS, RS, FOLD>(t.gout[k] ? t.gout[k] : gout, rois, t.box[k], t.gfeat[k], n, B, C, t.H[k], t.W[k], ph, pw,
                                     t.scale[k], sampling_ratio, aligned, t.accumulate[k], chunks, t.segs[k], t.hts[k],
                                     task - t.task0[k], active, t.tbuf[k]);
}

// footprint boxes of every RoI on every level slot of the table (blockIdx.y = slot; b = -1 where the RoI is on another level)
__global__ __launch_bounds__(256) void roi_bbox_levels_kernel(const float *__restrict__ rois, const int64_t *__restrict__ roi_level,
                                                              GatherLevels t, int64_t n, int B, int ph, int pw, int sampling_ratio,
                                                              int aligned)
{
    const int64_t ri = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ri >= n) return;
    const int k = blockIdx.y;
    const int H = t.H[k], W = t.W[k];
    RoiBox o{};
    o.b = -1;
    if (!roi_level || roi_level[ri] == (int64_t)t.level[k]) {       // no level list: every RoI on every level (BA)
        const RoiGeom g = roi_geometry(rois + 5 * ri, t.scale[k], ph, pw, sampling_ratio, aligned);
        int r_lo, r_hi, c_lo, c_hi, t0, t1;
        axis_span(g.start_h, g.bin_h, 0, g.grid_h, H, r_lo, t0);
        axis_span(g.start_h, g.bin_h, ph - 1, g.grid_h, H, t1, r_hi);
        bool ok = g.batch >= 0 && g.batch < B && !(t0 < r_lo || r_hi < t1);
        axis_span(g.start_w, g.bin_w, 0, g.grid_w, W, c_lo, t0);
        axis_span(g.start_w, g.bin_w, pw - 1, g.grid_w, W, t1, c_hi);
        ok = ok && !(t0 < c_lo || c_hi < t1);
        if (ok) { o.b = (short)g.batch; o.r_lo = (short)r_lo; o.r_hi = (short)r_hi; o.c_lo = (short)c_lo; o.c_hi = (short)c_hi; }
    }
    const_cast<RoiBox *>(t.box[k])[ri] = o;
}

// Wavefronts per tile of the gather backward (gather_tile RS): HTD_ROI_BWD_SPLIT = 1 / 2 / 4 forces it.  Measured
// (tools/bench_roi_align.py, B = 4 @ 800x1344): 2 048 RoIs, each on ONE level, all levels in one launch: 462 / 666 / 1 055 us
// for 1 / 2 / 4 -- the chains are short and the split costs registers (96 -> 128 VGPRs) and a barrier; 512 RoIs on EVERY level
// (BA), one launch: 1 220 / 1 146 / 1 213 us, launched level by level 2 309 / 1 694 / 1 543 us (the coarse maps' strips walk
// all RoIs of their image: the split shortens exactly that chain).  So: 2 when every RoI sits on every level, else 1.
int gather_split(int64_t n, int64_t tiles_min, bool every_level)
{
    static const int forced = getenv("HTD_ROI_BWD_SPLIT") ? atoi(getenv("HTD_ROI_BWD_SPLIT")) : 0;
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    (void)n; (void)tiles_min;
    return every_level ? 2 : 1;
}

int launch(bool backward, const float *in, const float *rois, const int64_t *roi_level, int level, float *out,
           int64_t n, int B, int C, int H, int W, int ph, int pw, float scale, int sr, int aligned,
           void *stream)
{
    HTD_REQUIRE(n >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0,
                "roi_align: bad sizes n=%lld B=%d C=%d H=%d W=%d", (long long)n, B, C, H, W);
    HTD_REQUIRE(C % 4 == 0, "roi_align: C=%d must be a multiple of 4 (float4 channel vectors)", C);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(in && rois && out, "roi_align: null pointer");
    const int chunks = (C + 255) / 256;
    const int waves_per_block = 4;
    hipStream_t s = (hipStream_t)stream;
    // few RoIs (BA: two dozen large ones on every level): the bin-wise kernel has 49 waves per RoI to spread a large
    // footprint over; many RoIs: the row-wise kernel halves the atomics
    if (backward && ph <= MAXP && pw <= MAXP && n * chunks >= 256) {
        const int row_slots = (int)std::min<int64_t>(H, std::max<int64_t>(8, htd::ceil_div(16384, n * chunks)));   // >= 16k waves
        const int64_t tasks = n * row_slots * chunks;
        const int64_t blocks = htd::ceil_div(tasks, waves_per_block);
        HTD_REQUIRE(blocks < (1ll << 31), "roi_align: too many tasks");
        hipLaunchKernelGGL(roi_align_bwd_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, rois, roi_level,
                           level, out, n, B, C, H, W, ph, pw, scale, sr, aligned, chunks, row_slots, tasks);
        return htd::check_launch("roi_align");
    }
    const int64_t tasks = n * ph * pw * chunks;
    if (!backward && tasks < 8192) {        // too few bins to fill the chip with one wavefront each: a workgroup per bin
        hipLaunchKernelGGL(roi_align_fwd_split_kernel<8>, dim3((unsigned)tasks), dim3(512), 0, s, in, rois, roi_level, level, out, B,
                           C, H, W, ph, pw, scale, sr, aligned, chunks);
        return htd::check_launch("roi_align");
    }
    const int64_t blocks = htd::ceil_div(tasks, waves_per_block);
    HTD_REQUIRE(blocks < (1ll << 31), "roi_align: too many tasks");
    if (backward)
        hipLaunchKernelGGL(roi_align_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, in, rois, roi_level, level,
                           out, n, B, C, H, W, ph, pw, scale, sr, aligned, chunks, tasks);
    else
        hipLaunchKernelGGL(roi_align_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, in, rois, roi_level, level,
                           out, n, B, C, H, W, ph, pw, scale, sr, aligned, chunks, tasks);
    return htd::check_launch("roi_align");
}

}  // namespace

extern "C" int htd_roi_align_fwd(const float *feat, const float *rois, const int64_t *roi_level, int level,
                                 float *out, int64_t n, int B, int C, int H, int W, int ph, int pw, float spatial_scale,
                                 int sampling_ratio, int aligned, void *stream)
{
    return launch(false, feat, rois, roi_level, level, out, n, B, C, H, W, ph, pw, spatial_scale, sampling_ratio,
                  aligned, stream);
}

// feats[l] [B][H[l]][W[l]][C] for l < L (L <= 8); every RoI is pooled from level roi_level[i] (RoIs with a level outside
// [0, L) get zeros): SingleRoIExtractor.forward (single_level_roi_extractor.py:81-99) in one launch.
static int roi_align_levels_fwd_impl(const float *const *feats, const int *H, const int *W, const float *scales, int L,
                                     const float *rois, const int64_t *roi_level, float *out, int64_t n, int B, int C, int ph,
                                     int pw, int sampling_ratio, int aligned, float *amax_out, void *stream)
{
    HTD_REQUIRE(L > 0 && L <= 8 && n >= 0 && B > 0 && C > 0 && ph > 0 && pw > 0, "roi_align_levels: bad sizes");
    HTD_REQUIRE(C % 4 == 0, "roi_align_levels: C=%d must be a multiple of 4", C);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(feats && H && W && scales && rois && roi_level && out, "roi_align_levels: null pointer");
    LevelTable tab{};
    for (int l = 0; l < L; ++l) {
        HTD_REQUIRE(feats[l] && H[l] > 0 && W[l] > 0, "roi_align_levels: bad level %d", l);
        tab.feat[l] = feats[l]; tab.H[l] = H[l]; tab.W[l] = W[l]; tab.scale[l] = scales[l];
    }
    const int chunks = (C + 255) / 256;
    const int64_t tasks = n * ph * pw * chunks;
    const int64_t blocks = htd::ceil_div(tasks, 4);
    HTD_REQUIRE(blocks < (1ll << 31), "roi_align_levels: too many tasks");
    hipLaunchKernelGGL(roi_align_levels_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab, rois, roi_level, out,
                       n, B, C, L, ph, pw, sampling_ratio, aligned, chunks, tasks, amax_out);
    return htd::check_launch("roi_align_levels");
}

extern "C" int htd_roi_align_levels_fwd(const float *const *feats, const int *H, const int *W, const float *scales, int L,
                                        const float *rois, const int64_t *roi_level, float *out, int64_t n, int B, int C, int ph,
                                        int pw, int sampling_ratio, int aligned, void *stream)
{
    return roi_align_levels_fwd_impl(feats, H, W, scales, L, rois, roi_level, out, n, B, C, ph, pw, sampling_ratio, aligned, nullptr,
                                     stream);
}

// the same, and max |out| is left in *amax_out (a device scalar holding zero or an earlier maximum on entry): the pooled tiles go
// straight into the RoI heads' first FC layer, whose H2 launches (htd_conv2d_fwd_x3h, htd_conv2d_bwd_weight_h2) scale by it
extern "C" int htd_roi_align_levels_fwd_amax(const float *const *feats, const int *H, const int *W, const float *scales, int L,
                                             const float *rois, const int64_t *roi_level, float *out, int64_t n, int B, int C,
                                             int ph, int pw, int sampling_ratio, int aligned, float *amax_out, void *stream)
{
    HTD_REQUIRE(amax_out, "roi_align_levels: null maximum");
    return roi_align_levels_fwd_impl(feats, H, W, scales, L, rois, roi_level, out, n, B, C, ph, pw, sampling_ratio, aligned, amax_out,
                                     stream);
}

// Every RoI pooled from every level (AdptRoIExtractor): outs[l] (n, ph, pw, C) <- RoIAlign(feats[l], rois), l < L, ONE launch.
// The same arithmetic as L calls of htd_roi_align_fwd (bit-identical results).
extern "C" int htd_roi_align_all_levels_fwd(const float *const *feats, const int *H, const int *W, const float *scales, int L,
                                            const float *rois, float *const *outs, int64_t n, int B, int C, int ph, int pw,
                                            int sampling_ratio, int aligned, void *stream)
{
    HTD_REQUIRE(L > 0 && L <= 8 && n >= 0 && B > 0 && C > 0 && ph > 0 && pw > 0, "roi_align_all_levels: bad sizes");
    HTD_REQUIRE(C % 4 == 0, "roi_align_all_levels: C=%d must be a multiple of 4", C);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(feats && H && W && scales && rois && outs, "roi_align_all_levels: null pointer");
    LevelTable tab{};
    AllLevelsOut o{};
    for (int l = 0; l < L; ++l) {
        HTD_REQUIRE(feats[l] && outs[l] && H[l] > 0 && W[l] > 0, "roi_align_all_levels: bad level %d", l);
        tab.feat[l] = feats[l]; tab.H[l] = H[l]; tab.W[l] = W[l]; tab.scale[l] = scales[l];
        o.out[l] = outs[l];
    }
    const int chunks = (C + 255) / 256;
    const int64_t tasks = (int64_t)L * n * ph * pw * chunks;
    HTD_REQUIRE(tasks < (1ll << 31), "roi_align_all_levels: too many tasks");
    hipStream_t s = (hipStream_t)stream;
    if (tasks < 4 * 8192)           // few bins (as htd_roi_align_fwd per level): a workgroup per bin
        hipLaunchKernelGGL(roi_align_all_levels_fwd_kernel<8>, dim3((unsigned)tasks), dim3(512), 0, s, tab, o, rois, n, B, C, L, ph, pw,
                           sampling_ratio, aligned, chunks, tasks);
    else
        hipLaunchKernelGGL(roi_align_all_levels_fwd_kernel<1>, dim3((unsigned)htd::ceil_div(tasks, 4)), dim3(256), 0, s, tab, o, rois, n,
                           B, C, L, ph, pw, sampling_ratio, aligned, chunks, tasks);
    return htd::check_launch("roi_align_all_levels");
}

extern "C" int64_t htd_roi_align_bwd_gather_workspace_bytes(int64_t n) { return (n > 0 ? n : 1) * (int64_t)sizeof(RoiBox); }

// Gather-form backward (no atomics, bit-stable): grad_feat = (accumulate ? grad_feat : 0) + RoIAlign^T(grad_out).
// Every pixel of grad_feat is written when accumulate == 0 (no memset needed).  workspace: ..._workspace_bytes(n).
extern "C" int htd_roi_align_bwd_gather(const float *grad_out, const float *rois, const int64_t *roi_level, int level,
                                        float *grad_feat, int64_t n, int B, int C, int H, int W, int ph, int pw,
                                        float spatial_scale, int sampling_ratio, int aligned, int accumulate, void *workspace,
                                        void *stream)
{
    HTD_REQUIRE(n >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && ph > 0 && pw > 0 && ph <= MAXP && pw <= MAXP,
                "roi_align_bwd_gather: bad sizes n=%lld B=%d C=%d H=%d W=%d out=%dx%d", (long long)n, B, C, H, W, ph, pw);
    HTD_REQUIRE(C % 4 == 0 && H < 32768 && W < 32768 && B < 32768, "roi_align_bwd_gather: C=%d must be a multiple of 4", C);
    HTD_REQUIRE(grad_feat && (n == 0 || (grad_out && rois && workspace)), "roi_align_bwd_gather: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (!accumulate && hipMemsetAsync(grad_feat, 0, (size_t)B * H * W * C * 4, s) != hipSuccess) {
            htd::set_error("roi_align_bwd_gather: memset failed");
            return HTD_ERR_LAUNCH;
        }
        return HTD_OK;
    }
    RoiBox *box = (RoiBox *)workspace;
    hipLaunchKernelGGL(roi_bbox_kernel, dim3((unsigned)htd::ceil_div(n, 256)), dim3(256), 0, s, rois, roi_level, level, box, n, B,
                       H, W, ph, pw, spatial_scale, sampling_ratio, aligned);
    const int segs = (W + GW_TILE - 1) / GW_TILE;
    // one-row strips of 8 pixels, float4 lanes (256-channel chunks).  Measured on the P2..P5 maps of B = 4 @ 800x1344 with
    // 2048 RoIs: 0.70 ms (96 VGPRs, 5 waves per SIMD) against 0.89 ms for 16-pixel strips (130 VGPRs), 1.08 ms for 4-row x
    // 128-channel tiles (fewer gout reads, but 225 VGPRs = 2 waves per SIMD) and 0.87 ms for the atomic scatter: the kernel
    // is bound by the chain of dependent loads per RoI, i.e. by how many wavefronts are in flight, not by bytes
    constexpr int rows = 1;
    const int chunks = (C + 255) / 256;
    const int hts = (H + rows - 1) / rows;
    const int64_t tasks = (int64_t)B * hts * segs * chunks;
    const int64_t blocks = htd::ceil_div(tasks, 4);
    HTD_REQUIRE(blocks < (1ll << 31), "roi_align_bwd_gather: too many tiles");
    const int rs = gather_split(n, tasks, roi_level == nullptr);
    if (rs == 4)
        hipLaunchKernelGGL((roi_align_bwd_gather_kernel<4, rows, 4>), dim3((unsigned)tasks), dim3(256), 0, s, grad_out, rois, box,
                           grad_feat, n, B, C, H, W, ph, pw, spatial_scale, sampling_ratio, aligned, accumulate, chunks, segs, hts, tasks);
    else if (rs == 2)
        hipLaunchKernelGGL((roi_align_bwd_gather_kernel<4, rows, 2>), dim3((unsigned)htd::ceil_div(tasks, 2)), dim3(256), 0, s, grad_out,
                           rois, box, grad_feat, n, B, C, H, W, ph, pw, spatial_scale, sampling_ratio, aligned, accumulate, chunks, segs,
                           hts, tasks);
    else
        hipLaunchKernelGGL((roi_align_bwd_gather_kernel<4, rows, 1>), dim3((unsigned)blocks), dim3(256), 0, s, grad_out, rois, box,
                           grad_feat, n, B, C, H, W, ph, pw, spatial_scale, sampling_ratio, aligned, accumulate, chunks, segs, hts, tasks);
    return htd::check_launch("roi_align_bwd_gather");
}

// Gather-form backward of a multi-level extractor, all levels in one launch:
//   grad_feats[l] = (accumulate[l] ? grad_feats[l] : 0) + RoIAlign_l^T(rows of level l);  grad_feats[l] == NULL skips level l.
// roi_level != NULL (SingleRoIExtractor): every RoI is pooled on the level roi_level names, one grad_out for all.
// roi_level == NULL (AdptRoIExtractor, adaptative_roi_extractor.py:66-76): every RoI on EVERY level, grad_outs[l] per level.
namespace {
int levels_bwd_gather(const float *grad_out, const float *const *grad_outs, const float *rois, const int64_t *roi_level,
                      float *const *grad_feats, const int *H, const int *W, const float *scales, const int *accumulate, int L,
                      int64_t n, int B, int C, int ph, int pw, int sampling_ratio, int aligned, void *workspace, void *stream,
                      const char *what, void *fold_ws = nullptr)
{
    HTD_REQUIRE(L > 0 && L <= GL_MAX && n >= 0 && B > 0 && C > 0 && ph > 0 && pw > 0 && ph <= MAXP && pw <= MAXP,
                "%s: bad sizes L=%d n=%lld B=%d C=%d out=%dx%d", what, L, (long long)n, B, C, ph, pw);
    HTD_REQUIRE(C % 4 == 0 && B < 32768 && grad_feats && H && W && scales && accumulate,
                "%s: C=%d must be a multiple of 4, tables non-null", what, C);
    HTD_REQUIRE(n == 0 || ((grad_out || grad_outs) && rois && workspace), "%s: null pointer", what);
    hipStream_t s = (hipStream_t)stream;
    GatherLevels t{};
    const int chunks = (C + 255) / 256;
    int64_t tasks = 0, tiles_min = 1ll << 62;
    for (int l = L - 1; l >= 0; --l) {                     // coarsest level first: its strips have the longest RoI lists
        if (!grad_feats[l]) continue;
        HTD_REQUIRE(H[l] > 0 && W[l] > 0 && H[l] < 32768 && W[l] < 32768, "%s: bad map size", what);
        if (n == 0) {
            if (!accumulate[l] && hipMemsetAsync(grad_feats[l], 0, (size_t)B * H[l] * W[l] * C * 4, s) != hipSuccess) {
                htd::set_error("%s: memset failed", what);
                return HTD_ERR_LAUNCH;
            }
            continue;
        }
        const int k = t.slots++;
        t.gfeat[k] = grad_feats[l];
        t.gout[k] = grad_outs ? grad_outs[l] : nullptr;
        HTD_REQUIRE(!grad_outs || grad_outs[l], "%s: level %d has a gradient map but no grad_out", what, l);
        t.box[k] = (const RoiBox *)workspace + (int64_t)l * n;
        t.H[k] = H[l]; t.W[k] = W[l]; t.scale[k] = scales[l]; t.accumulate[k] = accumulate[l]; t.level[k] = l;
        t.segs[k] = (W[l] + GW_TILE - 1) / GW_TILE;
        t.hts[k] = H[l];
        t.task0[k] = tasks;
        const int64_t tiles = (int64_t)B * t.hts[k] * t.segs[k] * chunks;
        tiles_min = std::min(tiles_min, tiles);
        tasks += tiles;
    }
    if (t.slots == 0) return HTD_OK;
    t.task0[t.slots] = tasks;
    int64_t ftasks = 0, foff = 0;            // folded bins: slot k's buffer behind the previous ones
    for (int k = 0; k < t.slots; ++k) {
        t.tbuf[k] = fold_ws ? (float *)fold_ws + foff : nullptr;
        foff += n * (int64_t)t.H[k] * pw * C;
        t.ftask0[k] = ftasks;
        ftasks += n * (int64_t)t.H[k] * chunks;
    }
    t.ftask0[t.slots] = ftasks;
    hipLaunchKernelGGL(roi_bbox_levels_kernel, dim3((unsigned)htd::ceil_div(n, 256), (unsigned)t.slots), dim3(256), 0, s, rois,
                       roi_level, t, n, B, ph, pw, sampling_ratio, aligned);
    HTD_REQUIRE(tasks < (1ll << 31), "%s: too many tiles", what);
    const int rs = gather_split(n, tiles_min, roi_level == nullptr);
    if (fold_ws) {
        HTD_REQUIRE(ftasks < (1ll << 33), "%s: too many fold tasks", what);
        hipLaunchKernelGGL(roi_fold_kernel, dim3((unsigned)htd::ceil_div(ftasks, 4)), dim3(256), 0, s, grad_out, rois, t, n, C, ph, pw,
                           sampling_ratio, aligned, chunks);
        if (rs == 4)
            hipLaunchKernelGGL((roi_align_bwd_gather_levels_kernel<4, 1, 4, true>), dim3((unsigned)tasks), dim3(256), 0, s, grad_out, rois,
                               t, n, B, C, ph, pw, sampling_ratio, aligned, chunks);
        else if (rs == 2)
            hipLaunchKernelGGL((roi_align_bwd_gather_levels_kernel<4, 1, 2, true>), dim3((unsigned)htd::ceil_div(tasks, 2)), dim3(256), 0,
                               s, grad_out, rois, t, n, B, C, ph, pw, sampling_ratio, aligned, chunks);
        else
            hipLaunchKernelGGL((roi_align_bwd_gather_levels_kernel<4, 1, 1, true>), dim3((unsigned)htd::ceil_div(tasks, 4)), dim3(256), 0,
                               s, grad_out, rois, t, n, B, C, ph, pw, sampling_ratio, aligned, chunks);
        return htd::check_launch(what);
    }
    if (rs == 4)
        hipLaunchKernelGGL((roi_align_bwd_gather_levels_kernel<4, 1, 4, false>), dim3((unsigned)tasks), dim3(256), 0, s, grad_out, rois, t,
                           n, B, C, ph, pw, sampling_ratio, aligned, chunks);
    else if (rs == 2)
        hipLaunchKernelGGL((roi_align_bwd_gather_levels_kernel<4, 1, 2, false>), dim3((unsigned)htd::ceil_div(tasks, 2)), dim3(256), 0, s,
                           grad_out, rois, t, n, B, C, ph, pw, sampling_ratio, aligned, chunks);
    else
        hipLaunchKernelGGL((roi_align_bwd_gather_levels_kernel<4, 1, 1, false>), dim3((unsigned)htd::ceil_div(tasks, 4)), dim3(256), 0, s,
                           grad_out, rois, t, n, B, C, ph, pw, sampling_ratio, aligned, chunks);
    return htd::check_launch(what);
}
}  // namespace

extern "C" int htd_roi_align_levels_bwd_gather(const float *grad_out, const float *rois, const int64_t *roi_level,
                                               float *const *grad_feats, const int *H, const int *W, const float *scales,
                                               const int *accumulate, int L, int64_t n, int B, int C, int ph, int pw,
                                               int sampling_ratio, int aligned, void *workspace, void *stream)
{
    HTD_REQUIRE(n == 0 || (grad_out && roi_level), "roi_align_levels_bwd_gather: null pointer");
    return levels_bwd_gather(grad_out, nullptr, rois, roi_level, grad_feats, H, W, scales, accumulate, L, n, B, C, ph, pw,
                             sampling_ratio, aligned, workspace, stream, "roi_align_levels_bwd_gather");
}

// every RoI pooled from EVERY level (BA): grad_outs[l] (n, ph, pw, C) is level l's pooled-feature gradient
extern "C" int htd_roi_align_all_levels_bwd_gather(const float *const *grad_outs, const float *rois, float *const *grad_feats,
                                                   const int *H, const int *W, const float *scales, const int *accumulate, int L,
                                                   int64_t n, int B, int C, int ph, int pw, int sampling_ratio, int aligned,
                                                   void *workspace, void *stream)
{
    HTD_REQUIRE(n == 0 || grad_outs, "roi_align_all_levels_bwd_gather: null pointer");
    return levels_bwd_gather(nullptr, grad_outs, rois, nullptr, grad_feats, H, W, scales, accumulate, L, n, B, C, ph, pw,
                             sampling_ratio, aligned, workspace, stream, "roi_align_all_levels_bwd_gather");
}

// The same two entry points with the bins folded along y once per (RoI, map row) in a pass of its own (roi_fold_kernel) instead of
// by every strip that the row crosses: bit-identical gradient maps, 2-4 x less L2 traffic in the gather.  fold_ws:
// htd_roi_align_fold_workspace_bytes(...) bytes; the buffers of the levels lie behind one another in launch order.
extern "C" int64_t htd_roi_align_fold_workspace_bytes(int64_t n, const int *H, int L, int pw, int C)
{
    int64_t rows = 0;
    for (int l = 0; l < L; ++l) rows += H[l];
    return (n > 0 ? n : 1) * rows * pw * C * 4;
}

extern "C" int htd_roi_align_levels_bwd_gather_folded(const float *grad_out, const float *rois, const int64_t *roi_level,
                                                      float *const *grad_feats, const int *H, const int *W, const float *scales,
                                                      const int *accumulate, int L, int64_t n, int B, int C, int ph, int pw,
                                                      int sampling_ratio, int aligned, void *workspace, void *fold_ws, void *stream)
{
    HTD_REQUIRE(n == 0 || (grad_out && roi_level && fold_ws), "roi_align_levels_bwd_gather_folded: null pointer");
    return levels_bwd_gather(grad_out, nullptr, rois, roi_level, grad_feats, H, W, scales, accumulate, L, n, B, C, ph, pw,
                             sampling_ratio, aligned, workspace, stream, "roi_align_levels_bwd_gather_folded", fold_ws);
}

extern "C" int htd_roi_align_all_levels_bwd_gather_folded(const float *const *grad_outs, const float *rois, float *const *grad_feats,
                                                          const int *H, const int *W, const float *scales, const int *accumulate,
                                                          int L, int64_t n, int B, int C, int ph, int pw, int sampling_ratio,
                                                          int aligned, void *workspace, void *fold_ws, void *stream)
{
    HTD_REQUIRE(n == 0 || (grad_outs && fold_ws), "roi_align_all_levels_bwd_gather_folded: null pointer");
    return levels_bwd_gather(nullptr, grad_outs, rois, nullptr, grad_feats, H, W, scales, accumulate, L, n, B, C, ph, pw,
                             sampling_ratio, aligned, workspace, stream, "roi_align_all_levels_bwd_gather_folded", fold_ws);
}

extern "C" int htd_roi_align_bwd(const float *grad_out, const float *rois, const int64_t *roi_level, int level,
                                 float *grad_feat, int64_t n, int B, int C, int H, int W, int ph, int pw,
                                 float spatial_scale, int sampling_ratio, int aligned, void *stream)
{
    return launch(true, grad_out, rois, roi_level, level, grad_feat, n, B, C, H, W, ph, pw, spatial_scale,
                  sampling_ratio, aligned, stream);
}
