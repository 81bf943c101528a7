// HBM-bound per-RoI operators of the HTD head on NHWC RoI tiles [n][P][C] (P = 7*7):
//   _fuse_global (+ alpha*enhanced), BA level fusion, global average pooling,
//   GroupNorm(+ReLU), and the SGD-momentum update of the flat parameter buffer.
// All are single-pass, float4-vectorised along C (16 B/lane => 1 KiB per wave access).
#include <algorithm>

#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// ------------------------------------------------------------------ fuse_global
__global__ __launch_bounds__(256) void fuse_global_fwd_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ rois,
                                                              const float *__restrict__ g,
                                                              const float *__restrict__ extra, float alpha,
                                                              float *__restrict__ out, int64_t total4, int P,
                                                              int C4, int B, float *__restrict__ amax_out)
{
    unsigned mx = 0u;                     // amax_out (may be NULL): max |out| of the launch, for an H2 consumer (common.h)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const int64_t i = idx / ((int64_t)P * C4);
        int b = (int)rois[5 * i];
        b = b < 0 ? 0 : (b >= B ? B - 1 : b);
        float4 v = ld4(x + idx * 4);
        const float4 gv = ld4(g + ((int64_t)b * C4 + c4) * 4);
        v.x += gv.x; v.y += gv.y; v.z += gv.z; v.w += gv.w;
        if (extra) {
            const float4 e = ld4(extra + idx * 4);
            v.x += alpha * e.x; v.y += alpha * e.y; v.z += alpha * e.z; v.w += alpha * e.w;
        }
        st4(out + idx * 4, v);
        mx = htd::mag_bits4(mx, v);
    }
    if (amax_out) htd::wave_mag_out(mx, amax_out);
}

// both[0 .. n) = x, both[n .. 2n) = x + g[img(roi)]: the two batches the HTD classification FCs run on, one read of x
__global__ __launch_bounds__(256) void plain_and_fused_kernel(const float *__restrict__ x, const float *__restrict__ rois,
                                                              const float *__restrict__ g, float *__restrict__ both,
                                                              int64_t total4, int P, int C4, int B, float *__restrict__ amax_out)
{
    unsigned mx = 0u;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const int64_t i = idx / ((int64_t)P * C4);
        int b = (int)rois[5 * i];
        b = b < 0 ? 0 : (b >= B ? B - 1 : b);
        float4 v = ld4(x + idx * 4);
        st4(both + idx * 4, v);
        mx = htd::mag_bits4(mx, v);
        const float4 gv = ld4(g + ((int64_t)b * C4 + c4) * 4);
        v.x += gv.x; v.y += gv.y; v.z += gv.z; v.w += gv.w;
        st4(both + (total4 + idx) * 4, v);
        mx = htd::mag_bits4(mx, v);
    }
    if (amax_out) htd::wave_mag_out(mx, amax_out);
}

// grad_global[b][c] += sum_{i in image b} sum_p grad[i][p][c]; a block walks ROIS_PER_BLOCK
// consecutive RoIs (RoIs arrive grouped by image, bbox2roi) and flushes once per image run.
constexpr int ROIS_PER_BLOCK = 16;
__global__ __launch_bounds__(256) void fuse_global_bwd_kernel(const float *__restrict__ grad,
                                                              const float *__restrict__ rois,
                                                              float *__restrict__ gg, int64_t n, int P, int C,
                                                              int B)
{
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t i0 = (int64_t)blockIdx.x * ROIS_PER_BLOCK;
    const int64_t i1 = min(n, i0 + ROIS_PER_BLOCK);
    float acc = 0.f;
    int cur = -1;
    for (int64_t i = i0; i < i1; ++i) {
        int b = (int)rois[5 * i];
        b = b < 0 ? 0 : (b >= B ? B - 1 : b);
        if (b != cur) {
            if (cur >= 0) atomicAdd(gg + (int64_t)cur * C + c, acc);
            acc = 0.f;
            cur = b;
        }
        const float *p = grad + (i * P) * C + c;
        for (int q = 0; q < P; ++q) acc += p[(int64_t)q * C];
    }
    if (cur >= 0) atomicAdd(gg + (int64_t)cur * C + c, acc);
}

// C % 4 == 0: a wave covers 256 channels with 16-byte loads, the four waves of a block take every fourth tile
// position, VROIS RoIs per block; partial sums meet in LDS and leave with one atomic per (image run, channel).
constexpr int VROIS = 4;
__global__ __launch_bounds__(256) void fuse_global_bwd_vec_kernel(const float *__restrict__ grad,
                                                                  const float *__restrict__ rois,
                                                                  float *__restrict__ gg, int64_t n, int P, int C4,
                                                                  int B)
{
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * VROIS, i1 = min(n, i0 + VROIS);
    for (int c4 = blockIdx.y * 64 + lane; c4 - lane < C4; c4 += gridDim.y * 64) {
        const bool live = c4 < C4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int cur = -1;
        for (int64_t i = i0; i <= i1; ++i) {
            int b = -1;
            if (i < i1) {
                b = (int)rois[5 * i];
                b = b < 0 ? 0 : (b >= B ? B - 1 : b);
            }
            if (b != cur) {                       // image run ends (uniform across the block): fold and flush
                if (cur >= 0) {
                    red[wave][lane] = acc;
                    __syncthreads();
                    if (wave == 0 && live) {
                        float4 t = red[0][lane];
                        for (int w = 1; w < 4; ++w) { t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w; }
                        float *dst = gg + ((int64_t)cur * C4 + c4) * 4;
                        atomicAdd(dst, t.x); atomicAdd(dst + 1, t.y); atomicAdd(dst + 2, t.z); atomicAdd(dst + 3, t.w);
                    }
                    __syncthreads();
                }
                acc = make_float4(0.f, 0.f, 0.f, 0.f);
                cur = b;
            }
            if (i < i1 && live) {
                const float4 *p = reinterpret_cast<const float4 *>(grad) + (i * P) * C4 + c4;
                for (int q = wave; q < P; q += 4) {
                    const float4 v = p[(int64_t)q * C4];
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ BA fusion
struct Ptr4 { const float *p[4]; };
struct MPtr4 { float *p[4]; };

__device__ __forceinline__ void softmax4(const float *att, int L, int64_t n, int64_t i, float *w)
{
    float m = -INFINITY;
    for (int l = 0; l < L; ++l) m = fmaxf(m, att[l * n + i]);
    float s = 0.f;
    for (int l = 0; l < L; ++l) { w[l] = __expf(att[l * n + i] - m); s += w[l]; }
    const float inv = 1.f / s;
    for (int l = 0; l < L; ++l) w[l] *= inv;
}

__device__ __forceinline__ bool on_ring(int p, int ph, int pw, int edge)
{
    const int y = p / pw, x = p % pw;
    return y < edge || y >= ph - edge || x < edge || x >= pw - edge;
}

__global__ __launch_bounds__(256) void ba_fuse_fwd_kernel(Ptr4 lvl, int L, const float *__restrict__ border,
                                                          const float *__restrict__ att, float *__restrict__ out,
                                                          int64_t n, int ph, int pw, int C4, int edge)
{
    const int64_t i = blockIdx.x;
    float w[4];
    softmax4(att, L, n, i, w);
    const int P = ph * pw;
    const int64_t base = i * P * C4;
    for (int e = threadIdx.x; e < P * C4; e += blockDim.x) {
        const int p = e / C4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int l = 0; l < L; ++l) {
            const float4 v = ld4(lvl.p[l] + (base + e) * 4);
            acc.x += w[l] * v.x; acc.y += w[l] * v.y; acc.z += w[l] * v.z; acc.w += w[l] * v.w;
        }
        if (on_ring(p, ph, pw, edge)) {
            const float4 v = ld4(border + (base + e) * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        st4(out + (base + e) * 4, acc);
    }
}

__global__ __launch_bounds__(256) void ba_fuse_bwd_kernel(Ptr4 lvl, int L, const float *__restrict__ att,
                                                          const float *__restrict__ go, MPtr4 glvl,
                                                          float *__restrict__ gborder, float *__restrict__ gatt,
                                                          int64_t n, int ph, int pw, int C4, int edge)
{
    const int64_t i = blockIdx.x;
    float w[4];
    softmax4(att, L, n, i, w);
    const int P = ph * pw;
    const int64_t base = i * P * C4;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int e = threadIdx.x; e < P * C4; e += blockDim.x) {
        const int p = e / C4;
        const float4 g = ld4(go + (base + e) * 4);
        const bool ring = on_ring(p, ph, pw, edge);
        for (int l = 0; l < L; ++l) {
            const float4 v = ld4(lvl.p[l] + (base + e) * 4);
            d[l] += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
            float4 o = make_float4(w[l] * g.x, w[l] * g.y, w[l] * g.z, w[l] * g.w);
            if (l == 0 && gborder == nullptr && ring) {       // the border source IS level 0: its share lands in the same map
                o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w;
            }
            st4(glvl.p[l] + (base + e) * 4, o);
        }
        if (gborder != nullptr) st4(gborder + (base + e) * 4, ring ? g : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int l = 0; l < 4; ++l) {
        const float s = htd::wave_sum(d[l]);
        if (lane == 0) red[wv][l] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float dl[4], dot = 0.f;
        for (int l = 0; l < L; ++l) {
            dl[l] = red[0][l] + red[1][l] + red[2][l] + red[3][l];
            dot += w[l] * dl[l];
        }
        for (int l = 0; l < L; ++l) gatt[l * n + i] = w[l] * (dl[l] - dot);
    }
}

// ------------------------------------------------------------------ global average pool
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float *__restrict__ x, float *__restrict__ out, int P,
                                                      int C)
{
    const int64_t b = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const float *p = x + b * P * C + c;
    float acc = 0.f;
    int q = 0;
    for (; q + 7 < P; q += 8) {              // eight loads in flight, added in position order (same sum, no chain of dependent loads)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(q + u) * C];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; q < P; ++q) acc += p[(int64_t)q * C];
    out[b * C + c] = acc / (float)P;
}

// acc != 0: gx += g / P (gx already holds another consumer's gradient of the pooled map: BA's level features feed the
// attention pooling AND the weighted sum -- the pooling's share is added in place instead of a second map that autograd sums)
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float *__restrict__ g, float *__restrict__ gx,
                                                      int64_t total4, int P, int C4, int acc)
{
    const float inv = 1.f / (float)P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const int64_t b = idx / ((int64_t)P * C4);
        const float4 v = ld4(g + (b * C4 + c4) * 4);
        float4 o = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
        if (acc) {
            const float4 t = ld4(gx + idx * 4);
            o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
        }
        st4(gx + idx * 4, o);
    }
}

// out_y[c] = sum over `rows` rows of ws[y * rows + r][c], y = blockIdx.y in {0, 1} (out0 / out1): few columns, many rows.  A
// workgroup takes 16 columns and spreads the rows over 16 thread groups, folded through LDS in a fixed order: deterministic.
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float *__restrict__ ws, float *__restrict__ out0,
                                                          float *__restrict__ out1, int n, int rows)
{
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + col;
    const float *src = ws + (int64_t)blockIdx.y * rows * n;
    float s = 0.f;
    if (c < n)
        for (int k = grp; k < rows; k += 16) s += src[(int64_t)k * n + c];
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && c < n) {
        float t = red[0][col];
#pragma unroll
        for (int g = 1; g < 16; ++g) t += red[g][col];
        (blockIdx.y ? out1 : out0)[c] = t;
    }
}

// Bit-reproducible form of fuse_global_bwd: (1) tile[i][c] = sum_p grad[i][p][c], one workgroup per RoI, the four waves take
// every fourth tile position and meet in LDS in a fixed order; (2) gg[b][c] = sum of tile[i][c] over the RoIs of image b in
// ascending i (the RoI list need not be grouped by image).  No float atomics: the gradient of the SFA feature -- and with it
// everything upstream of P6 -- is the same on every run.
__global__ __launch_bounds__(256) void roi_tile_sums_kernel(const float *__restrict__ grad, float *__restrict__ tile, int P,
                                                            int C4)
{
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = blockIdx.x;
    for (int c0 = 0; c0 < C4; c0 += 64) {
        const int c4 = c0 + lane;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < C4) {
            const float4 *p = reinterpret_cast<const float4 *>(grad) + (i * P) * C4 + c4;
            for (int q = wave; q < P; q += 4) {
                const float4 v = p[(int64_t)q * C4];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        red[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && c4 < C4) {
            float4 t = red[0][lane];
            for (int w = 1; w < 4; ++w) { t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w; }
            reinterpret_cast<float4 *>(tile)[i * C4 + c4] = t;
        }
        __syncthreads();
    }
}

// partial[k][b][c] = sum of tile[i][c] over the RoIs i of chunk k (IMG_CHUNK consecutive RoIs) that belong to image b, ascending i
constexpr int IMG_CHUNK = 64;
__global__ __launch_bounds__(64) void image_sums_kernel(const float *__restrict__ tile, const float *__restrict__ rois,
                                                        float *__restrict__ partial, int64_t n, int C4, int B)
{
    const int b = blockIdx.x, c4 = blockIdx.y * 64 + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.z * IMG_CHUNK, i1 = min(n, i0 + IMG_CHUNK);
    if (c4 >= C4) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = i0; i < i1; ++i) {
        int rb = (int)rois[5 * i];
        rb = rb < 0 ? 0 : (rb >= B ? B - 1 : rb);
        if (rb != b) continue;                      // uniform across the workgroup
        const float4 v = reinterpret_cast<const float4 *>(tile)[i * C4 + c4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4 *>(partial)[((int64_t)blockIdx.z * B + b) * C4 + c4] = acc;
}

// gg[e] = sum_k partial[k][e] in ascending k
__global__ __launch_bounds__(256) void chunk_sums_kernel(const float *__restrict__ partial, float *__restrict__ gg, int64_t total,
                                                         int chunks)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(int64_t)k * total + e];
    gg[e] = s;
}

// ------------------------------------------------------------------ GroupNorm (+ReLU)
// One block per RoI tile; thread = channel, so a group's `cpg` channels sit on adjacent
// lanes and the group statistics are a segmented wave reduction (cpg must divide 64).
// PREG > 0: the P <= PREG positions of a channel are read ONCE into registers (all loads in flight together) and the three
// passes -- sum, squared deviations, normalise -- run on them in the same order as the memory form: same bits, a third of
// the reads and no chain of dependent loads (the 7x7 tiles of the RoI regression branch: 3.2 -> 1.x ms for 32 768 RoIs).
template <int PREG>
__global__ void gn_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                              const float *__restrict__ beta, float *__restrict__ y, float *__restrict__ mean,
                              float *__restrict__ rstd, int P, int C, int G, float eps, int relu, float *__restrict__ amax_out)
{
    __shared__ unsigned mxs[16];
    float omax = 0.f;                     // largest |y| stored (amax_out: for an H2 consumer of y, conv_x3.hip)
    const int64_t i = blockIdx.x;
    const int c = threadIdx.x;
    const int cpg = C / G;
    const bool act = c < C;
    const float *xp = x + i * P * C + c;
    float xr[PREG > 0 ? PREG : 1];
    if constexpr (PREG > 0) {
#pragma unroll
        for (int q = 0; q < PREG; ++q) xr[q] = (act && q < P) ? xp[(int64_t)q * C] : 0.f;
    }
    float s = 0.f;
    if constexpr (PREG > 0) {
#pragma unroll
        for (int q = 0; q < PREG; ++q) if (q < P) s += xr[q];
    } else {
        if (act) for (int q = 0; q < P; ++q) s += xp[(int64_t)q * C];
    }
    for (int o = 1; o < cpg; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mu = s / (float)(P * cpg);
    float v = 0.f;
    if constexpr (PREG > 0) {
#pragma unroll
        for (int q = 0; q < PREG; ++q) if (q < P) { const float d = xr[q] - mu; v += d * d; }
        if (!act) v = 0.f;
    } else {
        if (act) for (int q = 0; q < P; ++q) { const float d = xp[(int64_t)q * C] - mu; v += d * d; }
    }
    for (int o = 1; o < cpg; o <<= 1) v += __shfl_xor(v, o, 64);
    const float rs = rsqrtf(v / (float)(P * cpg) + eps);
    if (act) {
    if (c % cpg == 0) { mean[i * G + c / cpg] = mu; rstd[i * G + c / cpg] = rs; }
    const float ga = gamma[c] * rs, be = beta[c] - mu * gamma[c] * rs;
    float *yp = y + i * P * C + c;
    if constexpr (PREG > 0) {
#pragma unroll
        for (int q = 0; q < PREG; ++q)
            if (q < P) {
                float t = xr[q] * ga + be;
                if (relu) t = fmaxf(t, 0.f);
                yp[(int64_t)q * C] = t;
                omax = (t != t) ? t : ((omax != omax) ? omax : fmaxf(omax, fabsf(t)));
            }
    } else {
        for (int q = 0; q < P; ++q) {
            float t = xp[(int64_t)q * C] * ga + be;
            if (relu) t = fmaxf(t, 0.f);
            yp[(int64_t)q * C] = t;
            omax = (t != t) ? t : ((omax != omax) ? omax : fmaxf(omax, fabsf(t)));
        }
    }
    }   // act
    if (amax_out != nullptr) {            // one atomic per tile (NaN sorts above everything)
        unsigned bits = (omax != omax) ? 0x7fc00000u : __float_as_uint(omax);
        for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o));
        if ((threadIdx.x & 63) == 0) mxs[threadIdx.x >> 6] = bits;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nw = (blockDim.x + 63) >> 6;
            for (int w = 1; w < nw; ++w) bits = max(bits, mxs[w]);
            if (bits != 0u) atomicMax(reinterpret_cast<unsigned *>(amax_out), bits);
        }
    }
}

// PREG > 0 (P <= PREG): the masked gradient and the centred input of a channel are read once and kept in registers for the
// second pass (same expressions in the same order: same bits)
template <int PREG>
__global__ void gn_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                              const float *__restrict__ gamma, const float *__restrict__ mean,
                              const float *__restrict__ rstd, const float *__restrict__ gy, float *__restrict__ gx,
                              float *__restrict__ ggamma, float *__restrict__ gbeta, int P, int C, int G, int relu,
                              float *__restrict__ partial)
{
    const int64_t i = blockIdx.x;
    const int c = threadIdx.x;
    const int cpg = C / G;
    const bool act = c < C;
    const int g = act ? c / cpg : 0;
    const float mu = mean[i * G + g], rs = rstd[i * G + g];
    const float ga = act ? gamma[c] : 0.f;
    const float *xp = x + i * P * C + c, *yp = y + i * P * C + c, *gp = gy + i * P * C + c;
    float sg = 0.f, sgx = 0.f;  // sum dy, sum dy*xhat for this channel
    float dr[PREG > 0 ? PREG : 1], xc[PREG > 0 ? PREG : 1];
    if constexpr (PREG > 0) {
        float yv[PREG];
#pragma unroll
        for (int q = 0; q < PREG; ++q) {
            const bool in = act && q < P;
            dr[q] = in ? gp[(int64_t)q * C] : 0.f;
            yv[q] = in ? yp[(int64_t)q * C] : 1.f;
            xc[q] = in ? xp[(int64_t)q * C] : mu;
        }
#pragma unroll
        for (int q = 0; q < PREG; ++q) {
            if (relu && !(yv[q] > 0.f)) dr[q] = 0.f;
            xc[q] = xc[q] - mu;
            if (act && q < P) {
                sg += dr[q];
                sgx += dr[q] * xc[q] * rs;
            }
        }
    } else if (act)
        for (int q = 0; q < P; ++q) {
            float d = gp[(int64_t)q * C];
            if (relu && !(yp[(int64_t)q * C] > 0.f)) d = 0.f;
            sg += d;
            sgx += d * (xp[(int64_t)q * C] - mu) * rs;
        }
    // partial != NULL: this tile's sums go to partial[tile][c] (gamma) and partial[n + tile][c] (beta), added up in a fixed
    // order by colsum_rows_kernel -- bit-reproducible; NULL: float atomics into ggamma / gbeta (htd_group_norm_relu_bwd)
    if (act) {
        if (partial) {
            partial[i * C + c] = sgx;
            partial[((int64_t)gridDim.x + i) * C + c] = sg;
        } else {
            atomicAdd(ggamma + c, sgx);
            atomicAdd(gbeta + c, sg);
        }
    }
    float a = sg * ga, b = sgx * ga;
    for (int o = 1; o < cpg; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    const float m = 1.f / (float)(P * cpg);
    a *= m; b *= m;
    if (!act) return;
    float *gxp = gx + i * P * C + c;
    if constexpr (PREG > 0) {
#pragma unroll
        for (int q = 0; q < PREG; ++q)
            if (q < P) {
                const float xh = xc[q] * rs;
                gxp[(int64_t)q * C] = rs * (dr[q] * ga - a - xh * b);
            }
    } else {
        for (int q = 0; q < P; ++q) {
            float d = gp[(int64_t)q * C];
            if (relu && !(yp[(int64_t)q * C] > 0.f)) d = 0.f;
            const float xh = (xp[(int64_t)q * C] - mu) * rs;
            gxp[(int64_t)q * C] = rs * (d * ga - a - xh * b);
        }
    }
}


// ---- GroupNorm backward of a tile, bandwidth form (round 4) ------------------------------------------------------------------
// gn_bwd_kernel gives a thread ONE channel and all P positions of it: 147 registers of cached operands at P = 49, nine waves per
// tile, one tile per CU in flight, 4-byte loads -- 0.31 TB/s on the regression branch's 58 MB tiles (561 us per launch in the
// trained-like step).  Here a thread is (four channels, every R-th position): 16-byte loads that run through the tile's
// contiguous [P][C] block, PQ <= 8 positions cached per thread, the per-channel sums of the R position groups meet in LDS in
// a fixed order, the group sums (cpg / 4 float4 columns) likewise.  Same formulas; the sums are taken in another (fixed) order.
// amax_out (may be NULL): the largest |gx| stored, for an H2 consumer (conv_x3.hip).
template <int PQ>
__global__ __launch_bounds__(1024) void gn_bwd_tile_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ gamma, const float *__restrict__ mean,
                                                           const float *__restrict__ rstd, const float *__restrict__ gy,
                                                           float *__restrict__ gx, float *__restrict__ ggamma,
                                                           float *__restrict__ gbeta, int P, int C, int G, int relu,
                                                           float *__restrict__ partial, int V, int R, float *__restrict__ amax_out)
{
    __shared__ float4 red[2][1024];
    __shared__ float grp[2][256];
    __shared__ unsigned mxs[16];
    const int64_t i = blockIdx.x;
    const int tid = threadIdx.x, c4 = tid % V, r = tid / V;
    const int cpg = C / G, cpg4 = cpg >> 2;
    const int g = (c4 * 4) / cpg;
    const float mu = mean[i * G + g], rs = rstd[i * G + g];
    const float4 ga = ld4(gamma + c4 * 4);
    const int64_t base = i * P * C + c4 * 4;
    float4 dr[PQ], xc[PQ];
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sgx = sg;
#pragma unroll
    for (int k = 0; k < PQ; ++k) {
        const int q = r + k * R;
        const bool in = q < P;
        const int64_t o = base + (int64_t)(in ? q : 0) * C;
        float4 d = ld4(gy + o), yv = ld4(y + o), xv = ld4(x + o);
        if (!in) d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (relu) {
            d.x = yv.x > 0.f ? d.x : 0.f; d.y = yv.y > 0.f ? d.y : 0.f; d.z = yv.z > 0.f ? d.z : 0.f; d.w = yv.w > 0.f ? d.w : 0.f;
        }
        xv = make_float4(xv.x - mu, xv.y - mu, xv.z - mu, xv.w - mu);
        dr[k] = d; xc[k] = xv;
        sg.x += d.x; sg.y += d.y; sg.z += d.z; sg.w += d.w;
        sgx.x += d.x * xv.x * rs; sgx.y += d.y * xv.y * rs; sgx.z += d.z * xv.z * rs; sgx.w += d.w * xv.w * rs;
    }
    red[0][r * V + c4] = sg;
    red[1][r * V + c4] = sgx;
    __syncthreads();
    if (r == 0) {
        float4 tg = red[0][c4], tx = red[1][c4];
        for (int rr = 1; rr < R; ++rr) {
            const float4 a = red[0][rr * V + c4], b = red[1][rr * V + c4];
            tg.x += a.x; tg.y += a.y; tg.z += a.z; tg.w += a.w;
            tx.x += b.x; tx.y += b.y; tx.z += b.z; tx.w += b.w;
        }
        // partial != NULL: this tile's sums go to partial[tile][c] (gamma) and partial[n + tile][c] (beta), added up in a fixed
        // order by colsum_rows_kernel; NULL: float atomics into ggamma / gbeta
        if (partial) {
            st4(partial + i * C + c4 * 4, tx);
            st4(partial + ((int64_t)gridDim.x + i) * C + c4 * 4, tg);
        } else {
            atomicAdd(ggamma + c4 * 4, tx.x); atomicAdd(ggamma + c4 * 4 + 1, tx.y); atomicAdd(ggamma + c4 * 4 + 2, tx.z); atomicAdd(ggamma + c4 * 4 + 3, tx.w);
            atomicAdd(gbeta + c4 * 4, tg.x); atomicAdd(gbeta + c4 * 4 + 1, tg.y); atomicAdd(gbeta + c4 * 4 + 2, tg.z); atomicAdd(gbeta + c4 * 4 + 3, tg.w);
        }
        grp[0][c4] = tg.x * ga.x + tg.y * ga.y + tg.z * ga.z + tg.w * ga.w;
        grp[1][c4] = tx.x * ga.x + tx.y * ga.y + tx.z * ga.z + tx.w * ga.w;
    }
    __syncthreads();
    float a = 0.f, b = 0.f;
    const int g0 = (c4 / cpg4) * cpg4;
    for (int j = 0; j < cpg4; ++j) { a += grp[0][g0 + j]; b += grp[1][g0 + j]; }
    const float m = 1.f / (float)(P * cpg);
    a *= m; b *= m;
    float omax = 0.f;
#pragma unroll
    for (int k = 0; k < PQ; ++k) {
        const int q = r + k * R;
        if (q < P) {
            float4 o;
            o.x = rs * (dr[k].x * ga.x - a - xc[k].x * rs * b);
            o.y = rs * (dr[k].y * ga.y - a - xc[k].y * rs * b);
            o.z = rs * (dr[k].z * ga.z - a - xc[k].z * rs * b);
            o.w = rs * (dr[k].w * ga.w - a - xc[k].w * rs * b);
            st4(gx + base + (int64_t)q * C, o);
            omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            if (o.x != o.x || o.y != o.y || o.z != o.z || o.w != o.w) omax = __uint_as_float(0x7fc00000u);
        }
    }
    if (amax_out != nullptr) {            // one atomic per tile (NaN sorts above everything)
        unsigned bits = (omax != omax) ? 0x7fc00000u : __float_as_uint(omax);
        for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o));
        if ((tid & 63) == 0) mxs[tid >> 6] = bits;
        __syncthreads();
        if (tid == 0) {
            const int nw = (blockDim.x + 63) >> 6;
            for (int w = 1; w < nw; ++w) bits = max(bits, mxs[w]);
            if (bits != 0u) atomicMax(reinterpret_cast<unsigned *>(amax_out), bits);
        }
    }
}

// host: -> true when the tile form took the launch
static bool gn_tile_off()
{
    static const bool off = getenv("HTD_GN_TILE") && atoi(getenv("HTD_GN_TILE")) == 0;
    return off;
}

static bool launch_gn_bwd_tile(const float *x, const float *y, const float *gamma, const float *mean, const float *rstd, const float *gy,
                               float *gx, float *ggamma, float *gbeta, int64_t n, int P, int C, int G, int relu, float *partial,
                               float *amax_out, hipStream_t s)
{
    const int cpg = C / G;
    if (gn_tile_off() || (C & 3) || (cpg & 3) || C / 4 > 256 || P < 1) return false;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gy | (uintptr_t)gx | (uintptr_t)gamma | (uintptr_t)partial) & 15) != 0) return false;
    const int V = C / 4;
    const int R = std::min(1024 / V, P);
    const int PQ = (int)htd::ceil_div(P, R);
    if (PQ > 8) return false;
    const dim3 grid((unsigned)n), block((unsigned)(R * V));
#define HTD_GN_TILE_CASE(K) \
    case K: hipLaunchKernelGGL(gn_bwd_tile_kernel<K>, grid, block, 0, s, x, y, gamma, mean, rstd, gy, gx, ggamma, gbeta, P, C, G, relu, partial, V, R, amax_out); break;
    switch (PQ) {
        HTD_GN_TILE_CASE(1) HTD_GN_TILE_CASE(2) HTD_GN_TILE_CASE(3) HTD_GN_TILE_CASE(4)
        HTD_GN_TILE_CASE(5) HTD_GN_TILE_CASE(6) HTD_GN_TILE_CASE(7) HTD_GN_TILE_CASE(8)
    }
#undef HTD_GN_TILE_CASE
    return true;
}

// ------------------------------------------------------------------ SGD momentum
__global__ __launch_bounds__(256) void sgd_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                  float *__restrict__ m, int64_t n, const float *__restrict__ lr_dev,
                                                  float mom, float wd, float gscale)
{
    const float lr = *lr_dev;
    const int64_t n4 = n >> 2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n4;
         idx += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = ld4(p + idx * 4), mv = ld4(m + idx * 4);
        const float4 gv = ld4(g + idx * 4);
        mv.x = mom * mv.x + (gv.x * gscale + wd * pv.x); pv.x -= lr * mv.x;
        mv.y = mom * mv.y + (gv.y * gscale + wd * pv.y); pv.y -= lr * mv.y;
        mv.z = mom * mv.z + (gv.z * gscale + wd * pv.z); pv.z -= lr * mv.z;
        mv.w = mom * mv.w + (gv.w * gscale + wd * pv.w); pv.w -= lr * mv.w;
        st4(p + idx * 4, pv);
        st4(m + idx * 4, mv);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) {
            m[i] = mom * m[i] + (g[i] * gscale + wd * p[i]);
            p[i] -= lr * m[i];
        }
}

// ------------------------------------------------------------------ frozen-statistics BatchNorm folding
// w'[co][k] = w[co][k] * s[co],  b'[co] = beta[co] - mean[co] * s[co],  s = gamma / sqrt(var + eps)
__global__ __launch_bounds__(256) void bn_fold_fwd_kernel(const float *__restrict__ w, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, const float *__restrict__ mean,
                                                          const float *__restrict__ var, float eps,
                                                          float *__restrict__ wf, float *__restrict__ bf, int K)
{
    const int co = blockIdx.x;
    const float s = gamma[co] * rsqrtf(var[co] + eps);
    if (threadIdx.x == 0) bf[co] = beta[co] - mean[co] * s;
    const float *wr = w + (int64_t)co * K;
    float *wo = wf + (int64_t)co * K;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        const float4 v = ld4(wr + k);
        st4(wo + k, make_float4(v.x * s, v.y * s, v.z * s, v.w * s));
    }
}

// given G = dL/dw' and gb = dL/db':  dw = G * s,  dbeta = gb,  dgamma = (sum_k G*w - mean*gb) / sqrt(var+eps)
__global__ __launch_bounds__(256) void bn_fold_bwd_kernel(const float *__restrict__ w, const float *__restrict__ gamma,
                                                          const float *__restrict__ mean, const float *__restrict__ var,
                                                          float eps, const float *__restrict__ gwf,
                                                          const float *__restrict__ gbf, float *__restrict__ gw,
                                                          float *__restrict__ ggamma, float *__restrict__ gbeta, int K)
{
    const int co = blockIdx.x;
    const float rs = rsqrtf(var[co] + eps);
    const float s = gamma[co] * rs;
    const float *wr = w + (int64_t)co * K, *gr = gwf + (int64_t)co * K;
    float *go = gw + (int64_t)co * K;
    float dot = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        const float4 v = ld4(wr + k), g = ld4(gr + k);
        dot += v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w;
        st4(go + k, make_float4(g.x * s, g.y * s, g.z * s, g.w * s));
    }
    __shared__ float red[4];
    dot = htd::wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = red[0] + red[1] + red[2] + red[3];
        const float gb = gbf[co];
        gbeta[co] = gb;
        ggamma[co] = (tot - mean[co] * gb) * rs;
    }
}

// ---- all frozen-BN folds (and the data-gradient weight images) of a ResNet stage in ONE launch ----------------------
// A stage of R101 has up to 70 conv + BN pairs; folded one by one that is a 5-microsecond launch per pair and pass, and
// every data gradient re-flips its weights (htd_conv2d_flip_weights) on top.  Here a device table describes the layers
// (pointers, shapes, prefix sums of their 32 x 32 tiles / of their rows) and a workgroup looks its layer up.
struct FoldDesc {
    const float *w, *gamma, *beta, *mean, *var;
    float *wf, *bf, *wT;            // wT (may be NULL): [ci][taps-1-t][co], the flipped / transposed image for dgrad
    int Co, Ci, taps, tile0;        // tile0: first tile of this layer in the launch
};
struct FoldBwdDesc {
    const float *w, *gamma, *mean, *var, *gwf, *gbf;
    float *gw, *ggamma, *gbeta;
    int Co, K, row0, pad;
};

template <typename D, typename F>
__device__ __forceinline__ int find_layer(const D *d, int n, int idx, F first)
{
    int lo = 0, hi = n - 1;                 // last layer whose first index is <= idx
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first(d[mid]) <= idx) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void bn_fold_many_fwd_kernel(const FoldDesc *__restrict__ descs, int n, float eps)
{
    __shared__ float tile[32][33];
    const int l = find_layer(descs, n, (int)blockIdx.x, [](const FoldDesc &x) { return x.tile0; });
    const FoldDesc d = descs[l];
    const int nco = (d.Co + 31) / 32, nci = (d.Ci + 31) / 32;
    int idx = (int)blockIdx.x - d.tile0;
    const int ci_t = idx % nci;
    idx /= nci;
    const int co_t = idx % nco, t = idx / nco;
    const int co0 = co_t * 32, ci0 = ci_t * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < d.Co && ci < d.Ci) {
            const int64_t i = ((int64_t)co * d.taps + t) * d.Ci + ci;
            if (d.gamma) {
                v = d.w[i] * (d.gamma[co] * rsqrtf(d.var[co] + eps));
                d.wf[i] = v;
            } else
                v = d.w[i];                  // no BN (gamma == NULL): the layer only wants its flipped image
        }
        tile[r][tx] = v;
    }
    if (d.gamma && t == 0 && ci_t == 0 && ty == 0 && co0 + tx < d.Co) {
        const int co = co0 + tx;
        const float s = d.gamma[co] * rsqrtf(d.var[co] + eps);
        d.bf[co] = d.beta[co] - d.mean[co] * s;
    }
    if (!d.wT) return;
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < d.Ci && co < d.Co) d.wT[((int64_t)ci * d.taps + (d.taps - 1 - t)) * d.Co + co] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void bn_fold_many_bwd_kernel(const FoldBwdDesc *__restrict__ descs, int n, float eps)
{
    const int l = find_layer(descs, n, (int)blockIdx.x, [](const FoldBwdDesc &x) { return x.row0; });
    const FoldBwdDesc d = descs[l];
    const int co = (int)blockIdx.x - d.row0, K = d.K;
    const float rs = rsqrtf(d.var[co] + eps);
    const float s = d.gamma[co] * rs;
    const float *wr = d.w + (int64_t)co * K, *gr = d.gwf + (int64_t)co * K;
    float *go = d.gw + (int64_t)co * K;
    float dot = 0.f;
    for (int k = threadIdx.x * 4; k < K; k += 1024) {
        const float4 v = ld4(wr + k), g = ld4(gr + k);
        dot += v.x * g.x + v.y * g.y + v.z * g.z + v.w * g.w;
        st4(go + k, make_float4(g.x * s, g.y * s, g.z * s, g.w * s));
    }
    __shared__ float red[4];
    dot = htd::wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = red[0] + red[1] + red[2] + red[3];
        const float gb = d.gbf[co];
        d.gbeta[co] = gb;
        d.ggamma[co] = (tot - d.mean[co] * gb) * rs;
    }
}

// out[i] = x[rows[i]] / gx[rows[i]] += g[i] for rows of F floats (F % 4 == 0), one float4 per thread: the stage-2 positives' rows of
// the RoI tiles (htd_roi_head.py:163-166 `bbox_feats[pos_inds]` and its adjoint; ATen's index_select / index_add_ ran these 26 MB
// at 0.1 TB/s)
__global__ __launch_bounds__(256) void rows_gather_kernel(const float4 *__restrict__ x, const int64_t *__restrict__ rows,
                                                          float4 *__restrict__ out, int64_t total, int F4)
{
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / F4;
        out[t] = x[rows[i] * F4 + (t - i * F4)];
    }
}
__global__ __launch_bounds__(256) void rows_add_kernel(const float4 *__restrict__ g, const int64_t *__restrict__ rows,
                                                       float4 *__restrict__ gx, int64_t total, int F4)
{
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / F4;
        const int64_t o = rows[i] * F4 + (t - i * F4);
        const float4 a = gx[o], b = g[t];
        gx[o] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

inline unsigned grid_for(int64_t work, int block = 256, int cap = 4096)
{
    int64_t b = htd::ceil_div(work, block);
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

static int fuse_global_fwd_impl(const float *roi_feats, const float *rois, const float *global_feat, const float *extra, float alpha,
                                float *out, int64_t n, int P, int C, int B, float *amax_out, void *stream)
{
    HTD_REQUIRE(C % 4 == 0 && P > 0 && B > 0 && n >= 0, "fuse_global: bad sizes n=%lld P=%d C=%d B=%d",
                (long long)n, P, C, B);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(roi_feats && rois && global_feat && out, "fuse_global: null pointer");
    const int64_t total4 = n * P * (C / 4);
    hipLaunchKernelGGL(fuse_global_fwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, roi_feats,
                       rois, global_feat, extra, alpha, out, total4, P, C / 4, B, amax_out);
    return htd::check_launch("fuse_global_fwd");
}

extern "C" int htd_fuse_global_fwd(const float *roi_feats, const float *rois, const float *global_feat,
                                   const float *extra, float alpha, float *out, int64_t n, int P, int C, int B,
                                   void *stream)
{
    return fuse_global_fwd_impl(roi_feats, rois, global_feat, extra, alpha, out, n, P, C, B, nullptr, stream);
}

// htd_fuse_global_fwd / htd_plain_and_fused_fwd that also leave the largest magnitude they store in *amax_out (a device scalar
// holding zero or an earlier maximum on entry): their outputs are what the RoI heads' first FC layers read, and those layers' H2
// launches (htd_conv2d_fwd_x3h, htd_conv2d_bwd_weight_h2) scale by it.
extern "C" int htd_fuse_global_fwd_amax(const float *roi_feats, const float *rois, const float *global_feat, const float *extra,
                                        float alpha, float *out, int64_t n, int P, int C, int B, float *amax_out, void *stream)
{
    HTD_REQUIRE(amax_out, "fuse_global: null maximum");
    return fuse_global_fwd_impl(roi_feats, rois, global_feat, extra, alpha, out, n, P, C, B, amax_out, stream);
}

// both [2n][P][C]: rows [0, n) = roi_feats, rows [n, 2n) = roi_feats + global_feat[image of the RoI] (htd_bbox_head.py:198,201)
static int plain_and_fused_impl(const float *roi_feats, const float *rois, const float *global_feat, float *both, int64_t n, int P,
                                int C, int B, float *amax_out, void *stream)
{
    HTD_REQUIRE(C % 4 == 0 && P > 0 && B > 0 && n >= 0, "plain_and_fused: bad sizes n=%lld P=%d C=%d B=%d", (long long)n, P, C, B);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(roi_feats && rois && global_feat && both, "plain_and_fused: null pointer");
    const int64_t total4 = n * P * (C / 4);
    hipLaunchKernelGGL(plain_and_fused_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, roi_feats, rois,
                       global_feat, both, total4, P, C / 4, B, amax_out);
    return htd::check_launch("plain_and_fused_fwd");
}

extern "C" int htd_plain_and_fused_fwd(const float *roi_feats, const float *rois, const float *global_feat, float *both,
                                       int64_t n, int P, int C, int B, void *stream)
{
    return plain_and_fused_impl(roi_feats, rois, global_feat, both, n, P, C, B, nullptr, stream);
}

extern "C" int htd_plain_and_fused_fwd_amax(const float *roi_feats, const float *rois, const float *global_feat, float *both,
                                            int64_t n, int P, int C, int B, float *amax_out, void *stream)
{
    HTD_REQUIRE(amax_out, "plain_and_fused: null maximum");
    return plain_and_fused_impl(roi_feats, rois, global_feat, both, n, P, C, B, amax_out, stream);
}

extern "C" int htd_fuse_global_bwd_global(const float *grad, const float *rois, float *grad_global, int64_t n, int P,
                                          int C, int B, void *stream)
{
    HTD_REQUIRE(P > 0 && C > 0 && B > 0 && n >= 0, "fuse_global_bwd: bad sizes");
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(grad && rois && grad_global, "fuse_global_bwd: null pointer");
    if ((C & 3) == 0 && (((uintptr_t)grad | (uintptr_t)grad_global) & 15) == 0) {
        dim3 grid((unsigned)htd::ceil_div(n, VROIS), (unsigned)htd::ceil_div(C / 4, 64));
        hipLaunchKernelGGL(fuse_global_bwd_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad, rois,
                           grad_global, n, P, C / 4, B);
        return htd::check_launch("fuse_global_bwd");
    }
    dim3 grid((unsigned)htd::ceil_div(n, ROIS_PER_BLOCK), (unsigned)htd::ceil_div(C, 256));
    hipLaunchKernelGGL(fuse_global_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad, rois, grad_global, n, P,
                       C, B);
    return htd::check_launch("fuse_global_bwd");
}

// Bit-reproducible form (no float atomics): grad_global [B][C] is OVERWRITTEN; workspace holds (n + ceil(n / 64) * B) * C
// floats; C % 4 == 0.
extern "C" int htd_fuse_global_bwd_global_ws(const float *grad, const float *rois, float *grad_global, int64_t n, int P,
                                             int C, int B, void *workspace, void *stream)
{
    HTD_REQUIRE(P > 0 && C > 0 && B > 0 && n >= 0 && (C & 3) == 0, "fuse_global_bwd_ws: bad sizes");
    HTD_REQUIRE(grad_global, "fuse_global_bwd_ws: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        hipMemsetAsync(grad_global, 0, (size_t)B * C * 4, s);
        return HTD_OK;
    }
    HTD_REQUIRE(grad && rois && workspace, "fuse_global_bwd_ws: null pointer");
    float *tile = (float *)workspace, *partial = tile + n * C;
    const int chunks = (int)htd::ceil_div(n, IMG_CHUNK);
    hipLaunchKernelGGL(roi_tile_sums_kernel, dim3((unsigned)n), dim3(256), 0, s, grad, tile, P, C / 4);
    hipLaunchKernelGGL(image_sums_kernel, dim3((unsigned)B, (unsigned)htd::ceil_div(C / 4, 64), (unsigned)chunks), dim3(64), 0, s,
                       (const float *)tile, rois, partial, n, C / 4, B);
    hipLaunchKernelGGL(chunk_sums_kernel, dim3((unsigned)htd::ceil_div((int64_t)B * C, 256)), dim3(256), 0, s,
                       (const float *)partial, grad_global, (int64_t)B * C, chunks);
    return htd::check_launch("fuse_global_bwd_ws");
}

extern "C" int htd_ba_fuse_fwd(const float *const *lvl, int L, const float *border, const float *att, float *out,
                               int64_t n, int ph, int pw, int C, int edge, void *stream)
{
    HTD_REQUIRE(L >= 1 && L <= 4, "ba_fuse: L=%d not in [1,4]", L);
    HTD_REQUIRE(C % 4 == 0 && ph > 0 && pw > 0 && edge >= 0, "ba_fuse: bad sizes");
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(lvl && border && att && out, "ba_fuse: null pointer");
    Ptr4 p{};
    for (int l = 0; l < L; ++l) p.p[l] = lvl[l];
    hipLaunchKernelGGL(ba_fuse_fwd_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, p, L, border, att,
                       out, n, ph, pw, C / 4, edge);
    return htd::check_launch("ba_fuse_fwd");
}

extern "C" int htd_ba_fuse_bwd(const float *const *lvl, int L, const float *att, const float *grad_out,
                               float *const *grad_lvl, float *grad_border, float *grad_att, int64_t n, int ph, int pw,
                               int C, int edge, void *stream)
{
    HTD_REQUIRE(L >= 1 && L <= 4, "ba_fuse: L=%d not in [1,4]", L);
    HTD_REQUIRE(C % 4 == 0 && ph > 0 && pw > 0 && edge >= 0, "ba_fuse: bad sizes");
    if (n == 0) return HTD_OK;
    // grad_border == NULL: the border source is level 0 (AdptRoIExtractor: the same RoIAlign output) and its gradient is added
    // into grad_lvl[0] by the kernel
    HTD_REQUIRE(lvl && att && grad_out && grad_lvl && grad_att, "ba_fuse_bwd: null pointer");
    Ptr4 p{};
    MPtr4 g{};
    for (int l = 0; l < L; ++l) { p.p[l] = lvl[l]; g.p[l] = grad_lvl[l]; }
    hipLaunchKernelGGL(ba_fuse_bwd_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, p, L, att, grad_out,
                       g, grad_border, grad_att, n, ph, pw, C / 4, edge);
    return htd::check_launch("ba_fuse_bwd");
}

extern "C" int htd_global_avg_pool_fwd(const float *x, float *out, int64_t B, int P, int C, void *stream)
{
    HTD_REQUIRE(P > 0 && C > 0 && B >= 0, "global_avg_pool: bad sizes");
    if (B == 0) return HTD_OK;
    HTD_REQUIRE(x && out, "global_avg_pool: null pointer");
    hipLaunchKernelGGL(gap_fwd_kernel, dim3((unsigned)B, (unsigned)htd::ceil_div(C, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, out, P, C);
    return htd::check_launch("global_avg_pool_fwd");
}

extern "C" int htd_global_avg_pool_bwd(const float *g, float *gx, int64_t B, int P, int C, void *stream)
{
    HTD_REQUIRE(P > 0 && C > 0 && C % 4 == 0 && B >= 0, "global_avg_pool_bwd: bad sizes");
    if (B == 0) return HTD_OK;
    HTD_REQUIRE(g && gx, "global_avg_pool_bwd: null pointer");
    const int64_t total4 = B * P * (C / 4);
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, g, gx, total4, P,
                       C / 4, 0);
    return htd::check_launch("global_avg_pool_bwd");
}

// gx [B][P][C] += g [B][C] / P
extern "C" int htd_global_avg_pool_bwd_acc(const float *g, float *gx, int64_t B, int P, int C, void *stream)
{
    HTD_REQUIRE(B >= 0 && P > 0 && C > 0 && C % 4 == 0, "global_avg_pool_bwd_acc: bad sizes");
    if (B == 0) return HTD_OK;
    HTD_REQUIRE(g && gx, "global_avg_pool_bwd_acc: null pointer");
    const int64_t total4 = B * P * (C / 4);
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, g, gx, total4, P, C / 4, 1);
    return htd::check_launch("global_avg_pool_bwd_acc");
}

static int gn_fwd_impl(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd, int64_t n, int P,
                       int C, int G, float eps, int relu, float *amax_out, void *stream)
{
    HTD_REQUIRE(G > 0 && C % G == 0, "group_norm: C=%d not divisible by G=%d", C, G);
    const int cpg = C / G;
    HTD_REQUIRE(cpg <= 64 && (cpg & (cpg - 1)) == 0, "group_norm: channels/group=%d must be a power of two <= 64", cpg);
    HTD_REQUIRE(C <= 1024, "group_norm: C=%d > 1024", C);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(x && gamma && beta && y && mean && rstd, "group_norm: null pointer");
    const int threads = (int)htd::ceil_div(C, 64) * 64;
    if (P <= 49)
        hipLaunchKernelGGL(gn_fwd_kernel<49>, dim3((unsigned)n), dim3(threads), 0, (hipStream_t)stream, x, gamma, beta, y,
                           mean, rstd, P, C, G, eps, relu, amax_out);
    else
        hipLaunchKernelGGL(gn_fwd_kernel<0>, dim3((unsigned)n), dim3(threads), 0, (hipStream_t)stream, x, gamma, beta, y,
                           mean, rstd, P, C, G, eps, relu, amax_out);
    return htd::check_launch("group_norm_fwd");
}

extern "C" int htd_group_norm_relu_fwd(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                                       float *rstd, int64_t n, int P, int C, int G, float eps, int relu, void *stream)
{
    return gn_fwd_impl(x, gamma, beta, y, mean, rstd, n, P, C, G, eps, relu, nullptr, stream);
}

// the same, and max |y| of what it stores is left in *amax_out (zero or an earlier maximum on entry): the `amax` of the
// convolution that consumes y on the H2 arithmetic (conv_x3.hip)
extern "C" int htd_group_norm_relu_fwd_amax(const float *x, const float *gamma, const float *beta, float *y, float *mean,
                                            float *rstd, int64_t n, int P, int C, int G, float eps, int relu, float *amax_out,
                                            void *stream)
{
    return gn_fwd_impl(x, gamma, beta, y, mean, rstd, n, P, C, G, eps, relu, amax_out, stream);
}

extern "C" int htd_group_norm_relu_bwd(const float *x, const float *y, const float *gamma, const float *mean,
                                       const float *rstd, const float *gy, float *gx, float *ggamma, float *gbeta,
                                       int64_t n, int P, int C, int G, int relu, void *stream)
{
    HTD_REQUIRE(G > 0 && C % G == 0, "group_norm: C=%d not divisible by G=%d", C, G);
    const int cpg = C / G;
    HTD_REQUIRE(cpg <= 64 && (cpg & (cpg - 1)) == 0, "group_norm: channels/group=%d must be a power of two <= 64", cpg);
    HTD_REQUIRE(C <= 1024, "group_norm: C=%d > 1024", C);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(x && y && gamma && mean && rstd && gy && gx && ggamma && gbeta, "group_norm_bwd: null pointer");
    const int threads = (int)htd::ceil_div(C, 64) * 64;
    if (launch_gn_bwd_tile(x, y, gamma, mean, rstd, gy, gx, ggamma, gbeta, n, P, C, G, relu, nullptr, nullptr, (hipStream_t)stream))
        return htd::check_launch("group_norm_bwd");
    hipLaunchKernelGGL((P <= 49 ? gn_bwd_kernel<49> : gn_bwd_kernel<0>), dim3((unsigned)n), dim3(threads), 0, (hipStream_t)stream, x, y, gamma, mean,
                       rstd, gy, gx, ggamma, gbeta, P, C, G, relu, (float *)nullptr);
    return htd::check_launch("group_norm_bwd");
}

// The same with bit-reproducible parameter gradients: per-tile sums go through `workspace` (2 * n * C floats) and are
// added in a fixed order; ggamma / gbeta are overwritten (no zero-initialisation needed).
static int gn_bwd_ws_impl(const float *x, const float *y, const float *gamma, const float *mean, const float *rstd, const float *gy,
                          float *gx, float *ggamma, float *gbeta, int64_t n, int P, int C, int G, int relu, void *workspace,
                          float *amax_out, void *stream)
{
    HTD_REQUIRE(G > 0 && C % G == 0, "group_norm: C=%d not divisible by G=%d", C, G);
    const int cpg = C / G;
    HTD_REQUIRE(cpg <= 64 && (cpg & (cpg - 1)) == 0, "group_norm: channels/group=%d must be a power of two <= 64", cpg);
    HTD_REQUIRE(C <= 1024, "group_norm: C=%d > 1024", C);
    HTD_REQUIRE(ggamma && gbeta, "group_norm_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        hipMemsetAsync(ggamma, 0, (size_t)C * 4, s);
        hipMemsetAsync(gbeta, 0, (size_t)C * 4, s);
        return HTD_OK;
    }
    HTD_REQUIRE(x && y && gamma && mean && rstd && gy && gx && workspace, "group_norm_bwd: null pointer");
    const int threads = (int)htd::ceil_div(C, 64) * 64;
    float *ws = (float *)workspace;
    if (!launch_gn_bwd_tile(x, y, gamma, mean, rstd, gy, gx, ggamma, gbeta, n, P, C, G, relu, ws, amax_out, s)) {
        HTD_REQUIRE(amax_out == nullptr, "group_norm_bwd: this shape (C=%d, P=%d) runs on the kernel that leaves no maximum", C, P);
        hipLaunchKernelGGL((P <= 49 ? gn_bwd_kernel<49> : gn_bwd_kernel<0>), dim3((unsigned)n), dim3(threads), 0, s, x, y, gamma, mean, rstd, gy, gx,
                           ggamma, gbeta, P, C, G, relu, ws);
    }
    hipLaunchKernelGGL(colsum_rows_kernel, dim3((unsigned)htd::ceil_div(C, 16), 2), dim3(256), 0, s, (const float *)ws, ggamma,
                       gbeta, C, (int)n);
    return htd::check_launch("group_norm_bwd_ws");
}

extern "C" int htd_group_norm_relu_bwd_ws(const float *x, const float *y, const float *gamma, const float *mean,
                                          const float *rstd, const float *gy, float *gx, float *ggamma, float *gbeta,
                                          int64_t n, int P, int C, int G, int relu, void *workspace, void *stream)
{
    return gn_bwd_ws_impl(x, y, gamma, mean, rstd, gy, gx, ggamma, gbeta, n, P, C, G, relu, workspace, nullptr, stream);
}

// 1 when htd_group_norm_relu_bwd_amax can leave the maximum of gx for this shape (the tile form of the kernel takes it)
extern "C" int htd_group_norm_bwd_amax_supported(int P, int C, int G)
{
    if (gn_tile_off() || G <= 0 || C % G != 0 || (C & 3) || ((C / G) & 3) || C / 4 > 256 || P < 1) return 0;
    const int V = C / 4, R = std::min(1024 / V, P);
    return htd::ceil_div(P, R) <= 8 ? 1 : 0;
}

// htd_group_norm_relu_bwd_ws that also leaves max |gx| in *amax_out (zero or an earlier maximum on entry): the `amax` of the
// convolution data gradient that consumes gx on the H2 arithmetic (conv_x3.hip)
extern "C" int htd_group_norm_relu_bwd_amax(const float *x, const float *y, const float *gamma, const float *mean,
                                            const float *rstd, const float *gy, float *gx, float *ggamma, float *gbeta,
                                            int64_t n, int P, int C, int G, int relu, void *workspace, float *amax_out, void *stream)
{
    return gn_bwd_ws_impl(x, y, gamma, mean, rstd, gy, gx, ggamma, gbeta, n, P, C, G, relu, workspace, amax_out, stream);
}

extern "C" int htd_sgd_momentum_step(float *param, const float *grad, float *momentum_buf, int64_t n,
                                     const float *lr_dev, float momentum, float weight_decay, float grad_scale,
                                     void *stream)
{
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(param && grad && momentum_buf && lr_dev, "sgd: null pointer");
    HTD_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15) == 0,
                "sgd: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       momentum_buf, n, lr_dev, momentum, weight_decay, grad_scale);
    return htd::check_launch("sgd");
}

extern "C" int htd_bn_fold_fwd(const float *w, const float *gamma, const float *beta, const float *mean,
                               const float *var, float eps, float *w_folded, float *b_folded, int Co, int K,
                               void *stream)
{
    HTD_REQUIRE(Co > 0 && K > 0 && K % 4 == 0, "bn_fold: bad sizes Co=%d K=%d", Co, K);
    HTD_REQUIRE(w && gamma && beta && mean && var && w_folded && b_folded, "bn_fold: null pointer");
    hipLaunchKernelGGL(bn_fold_fwd_kernel, dim3((unsigned)Co), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, mean,
                       var, eps, w_folded, b_folded, K);
    return htd::check_launch("bn_fold_fwd");
}

// desc: device array of n_layers FoldDesc { const float *w, *gamma, *beta, *mean, *var; float *wf, *bf, *wT; int Co, Ci,
// taps, tile0; } (80 bytes each); total_tiles = sum over layers of taps * ceil(Co/32) * ceil(Ci/32).
extern "C" int htd_bn_fold_many_fwd(const void *desc, int n_layers, int total_tiles, float eps, void *stream)
{
    static_assert(sizeof(FoldDesc) == 80, "FoldDesc layout is part of the ABI");
    HTD_REQUIRE(desc && n_layers > 0 && total_tiles > 0, "bn_fold_many_fwd: bad arguments");
    hipLaunchKernelGGL(bn_fold_many_fwd_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const FoldDesc *)desc, n_layers, eps);
    return htd::check_launch("bn_fold_many_fwd");
}

// desc: device array of n_layers FoldBwdDesc { const float *w, *gamma, *mean, *var, *gwf, *gbf; float *gw, *ggamma,
// *gbeta; int Co, K, row0, pad; } (88 bytes each); total_rows = sum of Co; K % 4 == 0.
extern "C" int htd_bn_fold_many_bwd(const void *desc, int n_layers, int total_rows, float eps, void *stream)
{
    static_assert(sizeof(FoldBwdDesc) == 88, "FoldBwdDesc layout is part of the ABI");
    HTD_REQUIRE(desc && n_layers > 0 && total_rows > 0, "bn_fold_many_bwd: bad arguments");
    hipLaunchKernelGGL(bn_fold_many_bwd_kernel, dim3((unsigned)total_rows), dim3(256), 0, (hipStream_t)stream,
                       (const FoldBwdDesc *)desc, n_layers, eps);
    return htd::check_launch("bn_fold_many_bwd");
}

extern "C" int htd_bn_fold_bwd(const float *w, const float *gamma, const float *mean, const float *var, float eps,
                               const float *gw_folded, const float *gb_folded, float *gw, float *ggamma, float *gbeta,
                               int Co, int K, void *stream)
{
    HTD_REQUIRE(Co > 0 && K > 0 && K % 4 == 0, "bn_fold: bad sizes Co=%d K=%d", Co, K);
    HTD_REQUIRE(w && gamma && mean && var && gw_folded && gb_folded && gw && ggamma && gbeta, "bn_fold_bwd: null pointer");
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3((unsigned)Co), dim3(256), 0, (hipStream_t)stream, w, gamma, mean, var,
                       eps, gw_folded, gb_folded, gw, ggamma, gbeta, K);
    return htd::check_launch("bn_fold_bwd");
}

// ------------------------------------------------------------------ max pooling (ResNet stem, NHWC)
// nn.MaxPool2d(kernel k, stride s, padding p) of backbones/resnet.py:509 on [B][H][W][C] maps: a thread owns four
// channels of one output pixel (16-byte accesses), padding = -inf.  idx (optional) records the flat input pixel
// (hi*W + wi) of the FIRST maximum in window scan order, which is where ATen sends the gradient.
namespace {

__global__ __launch_bounds__(256) void max_pool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                           int *__restrict__ idx, int B, int H, int W, int C4, int Ho,
                                                           int Wo, int k, int s, int p, float *__restrict__ amax_out)
{
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    unsigned mx = 0u;                     // amax_out (may be NULL): max |y| of the launch, for an H2 consumer (common.h)
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(t % C4);
        int64_t r = t / C4;
        const int wo = (int)(r % Wo);
        r /= Wo;
        const int ho = (int)(r % Ho), b = (int)(r / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int4 am = make_int4(-1, -1, -1, -1);
        for (int i = 0; i < k; ++i) {
            const int hi = ho * s - p + i;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int j = 0; j < k; ++j) {
                const int wi = wo * s - p + j;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 v = reinterpret_cast<const float4 *>(x)[(((int64_t)b * H + hi) * W + wi) * C4 + c4];
                const int pix = hi * W + wi;
                // NaN propagates like ATen's `(val > max) || isnan(val)`
                if (v.x > m.x || v.x != v.x) { m.x = v.x; am.x = pix; }
                if (v.y > m.y || v.y != v.y) { m.y = v.y; am.y = pix; }
                if (v.z > m.z || v.z != v.z) { m.z = v.z; am.z = pix; }
                if (v.w > m.w || v.w != v.w) { m.w = v.w; am.w = pix; }
            }
        }
        reinterpret_cast<float4 *>(y)[t] = m;
        if (idx) reinterpret_cast<int4 *>(idx)[t] = am;
        mx = htd::mag_bits4(mx, m);
    }
    if (amax_out) htd::wave_mag_out(mx, amax_out);
}

// gather form of the backward: an input pixel sums the gradients of the (at most ceil(k/s)^2) windows that contain
// it and elected it -- no atomics, bitwise reproducible.
__global__ __launch_bounds__(256) void max_pool_bwd_kernel(const float *__restrict__ g, const int *__restrict__ idx,
                                                           float *__restrict__ gx, int B, int H, int W, int C4, int Ho,
                                                           int Wo, int k, int s, int p)
{
    const int64_t total = (int64_t)B * H * W * C4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(t % C4);
        int64_t r = t / C4;
        const int wi = (int)(r % W);
        r /= W;
        const int hi = (int)(r % H), b = (int)(r / H);
        const int pix = hi * W + wi;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ho_lo = max(0, (hi + p - k + s) / s), ho_hi = min(Ho - 1, (hi + p) / s);
        const int wo_lo = max(0, (wi + p - k + s) / s), wo_hi = min(Wo - 1, (wi + p) / s);
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int64_t o = (((int64_t)b * Ho + ho) * Wo + wo) * C4 + c4;
                const int4 am = reinterpret_cast<const int4 *>(idx)[o];
                const float4 gv = reinterpret_cast<const float4 *>(g)[o];
                if (am.x == pix) acc.x += gv.x;
                if (am.y == pix) acc.y += gv.y;
                if (am.z == pix) acc.z += gv.z;
                if (am.w == pix) acc.w += gv.w;
            }
        reinterpret_cast<float4 *>(gx)[t] = acc;
    }
}

}  // namespace

namespace {
// Gradient of nearest-neighbour up-sampling (the FPN top-down sum, necks/fpn.py:177-186, fused into the lateral convolution's
// epilogue in the forward pass): out[b][rh][rw][:] = sum of g[b][ho][wo][:] over the fine pixels whose source is (rh, rw)
// under ATen's rule  source = min(floor(dst * in / out), in - 1).  One thread per coarse pixel and float4 of channels; the
// candidate fine rows / columns are tested with that very expression, so odd sizes agree with the forward bit for bit.
__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const float4 *__restrict__ g, float4 *__restrict__ out, int B,
                                                                   int H, int W, int h, int w, int C4, float sh, float sw)
{
    const int64_t total = (int64_t)B * h * w * C4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C4);
        int64_t t = e / C4;
        const int rw = (int)(t % w);
        t /= w;
        const int rh = (int)(t % h), b = (int)(t / h);
        const int ho0 = max(0, (int)floorf(rh / sh) - 1), ho1 = min(H - 1, (int)ceilf((rh + 1) / sh) + 1);
        const int wo0 = max(0, (int)floorf(rw / sw) - 1), wo1 = min(W - 1, (int)ceilf((rw + 1) / sw) + 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ho = ho0; ho <= ho1; ++ho) {
            if (min((int)floorf(ho * sh), h - 1) != rh) continue;
            for (int wo = wo0; wo <= wo1; ++wo) {
                if (min((int)floorf(wo * sw), w - 1) != rw) continue;
                const float4 v = g[(((int64_t)b * H + ho) * W + wo) * C4 + c];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        out[e] = acc;
    }
}
}  // namespace

namespace {
// The same on bf16 maps (four channels per thread, fp32 sums in the order of the fp32 kernel, one rounding at the end).
__global__ __launch_bounds__(256) void upsample_nearest_bwd_bf16_kernel(const uint2 *__restrict__ g, uint2 *__restrict__ out, int B,
                                                                        int H, int W, int h, int w, int C4, float sh, float sw)
{
    const int64_t total = (int64_t)B * h * w * C4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C4);
        int64_t t = e / C4;
        const int rw = (int)(t % w);
        t /= w;
        const int rh = (int)(t % h), b = (int)(t / h);
        const int ho0 = max(0, (int)floorf(rh / sh) - 1), ho1 = min(H - 1, (int)ceilf((rh + 1) / sh) + 1);
        const int wo0 = max(0, (int)floorf(rw / sw) - 1), wo1 = min(W - 1, (int)ceilf((rw + 1) / sw) + 1);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ho = ho0; ho <= ho1; ++ho) {
            if (min((int)floorf(ho * sh), h - 1) != rh) continue;
            for (int wo = wo0; wo <= wo1; ++wo) {
                if (min((int)floorf(wo * sw), w - 1) != rw) continue;
                const uint2 v = g[(((int64_t)b * H + ho) * W + wo) * C4 + c];
                acc[0] += __uint_as_float(v.x << 16); acc[1] += __uint_as_float(v.x & 0xffff0000u);
                acc[2] += __uint_as_float(v.y << 16); acc[3] += __uint_as_float(v.y & 0xffff0000u);
            }
        }
        unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {             // round to nearest even; NaN stays NaN
            const unsigned u = __float_as_uint(acc[k]);
            r[k] = (u & 0x7fffffffu) > 0x7f800000u ? ((u >> 16) | 0x40u) : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
        out[e] = make_uint2(r[0] | (r[1] << 16), r[2] | (r[3] << 16));
    }
}
}  // namespace

extern "C" int htd_upsample_nearest_bwd_bf16(const void *g, void *out, int B, int H, int W, int h, int w, int C, void *stream)
{
    HTD_REQUIRE(g && out && B > 0 && H > 0 && W > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0,
                "upsample_nearest_bwd_bf16: bad arguments");
    const int64_t total = (int64_t)B * h * w * (C / 4);
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 65536);
    hipLaunchKernelGGL(upsample_nearest_bwd_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint2 *)g,
                       (uint2 *)out, B, H, W, h, w, C / 4, (float)h / (float)H, (float)w / (float)W);
    return htd::check_launch("upsample_nearest_bwd_bf16");
}

// g [B][H][W][C] (fine) -> out [B][h][w][C] (coarse); replaces aten::upsample_nearest2d_backward on the FPN top-down path
extern "C" int htd_upsample_nearest_bwd(const float *g, float *out, int B, int H, int W, int h, int w, int C, void *stream)
{
    HTD_REQUIRE(g && out && B > 0 && H > 0 && W > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "upsample_nearest_bwd: bad arguments");
    const int64_t total = (int64_t)B * h * w * (C / 4);
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 65536);
    hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)g,
                       (float4 *)out, B, H, W, h, w, C / 4, (float)h / (float)H, (float)w / (float)W);
    return htd::check_launch("upsample_nearest_bwd");
}

static int max_pool2d_fwd_impl(const float *x, float *y, int *idx, int B, int H, int W, int C, int k, int stride, int pad,
                               float *amax_out, void *stream)
{
    HTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k,
                "max_pool2d: bad sizes");
    HTD_REQUIRE(C % 4 == 0, "max_pool2d: C=%d must be a multiple of 4", C);
    HTD_REQUIRE((int64_t)H * W < (1ll << 31), "max_pool2d: map too large");
    HTD_REQUIRE(x && y, "max_pool2d: null pointer");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    HTD_REQUIRE(Ho > 0 && Wo > 0, "max_pool2d: empty output");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 65536);
    hipLaunchKernelGGL(max_pool_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, idx, B, H, W, C / 4,
                       Ho, Wo, k, stride, pad, amax_out);
    return htd::check_launch("max_pool2d_fwd");
}

extern "C" int htd_max_pool2d_fwd(const float *x, float *y, int *idx, int B, int H, int W, int C, int k, int stride,
                                  int pad, void *stream)
{
    return max_pool2d_fwd_impl(x, y, idx, B, H, W, C, k, stride, pad, nullptr, stream);
}

// the same, and max |y| is left in *amax_out (device scalar, zero or an earlier maximum on entry) for the H2 launches of the 1x1
// layers that read the pooled map (htd_conv2d_fwd_x3h: the first block of the ResNet's first stage)
extern "C" int htd_max_pool2d_fwd_amax(const float *x, float *y, int *idx, int B, int H, int W, int C, int k, int stride, int pad,
                                       float *amax_out, void *stream)
{
    HTD_REQUIRE(amax_out, "max_pool2d: null maximum");
    return max_pool2d_fwd_impl(x, y, idx, B, H, W, C, k, stride, pad, amax_out, stream);
}

extern "C" int htd_max_pool2d_bwd(const float *g, const int *idx, float *gx, int B, int H, int W, int C, int k,
                                  int stride, int pad, void *stream)
{
    HTD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && C % 4 == 0,
                "max_pool2d_bwd: bad sizes");
    HTD_REQUIRE(g && idx && gx, "max_pool2d_bwd: null pointer");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int64_t total = (int64_t)B * H * W * (C / 4);
    const unsigned blocks = (unsigned)std::min<int64_t>(htd::ceil_div(total, 256), 65536);
    hipLaunchKernelGGL(max_pool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, idx, gx, B, H, W, C / 4,
                       Ho, Wo, k, stride, pad);
    return htd::check_launch("max_pool2d_bwd");
}

// Row selection and its adjoint: out[i][:] = x[rows[i]][:], i < n / gx[rows[i]][:] += g[i][:] (rows distinct: plain sums, bit-
// reproducible); rows of F floats, F % 4 == 0, every rows[i] in [0, N).  torch.index_select(x, 0, rows) / x.index_add_(0, rows, g).
extern "C" int htd_rows_gather(const float *x, const int64_t *rows, float *out, int64_t n, int64_t N, int F, void *stream)
{
    HTD_REQUIRE(n >= 0 && N >= 0 && F > 0 && F % 4 == 0, "rows_gather: bad sizes n=%lld N=%lld F=%d", (long long)n, (long long)N, F);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(x && rows && out, "rows_gather: null pointer");
    const int64_t total = n * (F / 4);
    hipLaunchKernelGGL(rows_gather_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, rows,
                       (float4 *)out, total, F / 4);
    return htd::check_launch("rows_gather");
}

extern "C" int htd_rows_add(const float *g, const int64_t *rows, float *gx, int64_t n, int64_t N, int F, void *stream)
{
    HTD_REQUIRE(n >= 0 && N >= 0 && F > 0 && F % 4 == 0, "rows_add: bad sizes n=%lld N=%lld F=%d", (long long)n, (long long)N, F);
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(g && rows && gx, "rows_add: null pointer");
    const int64_t total = n * (F / 4);
    hipLaunchKernelGGL(rows_add_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const float4 *)g, rows,
                       (float4 *)gx, total, F / 4);
    return htd::check_launch("rows_add");
}

