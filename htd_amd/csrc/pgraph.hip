// PGraph adjacency kernels (HTDBBoxHead.forward, roi_heads/bbox_heads/htd_bbox_head.py:198-219), batched over the
// (image, pyramid level) groups of a call: group g holds counts[g] RoIs in rows 0..counts[g]-1 of a [G][npad] padding.
//
//   pgraph_degree_kernel +     boxes -> A_local = D^-1/2 M D^-1/2,  M = (IoU with unit diagonal) > 0,  D = rowsum(M)
//   pgraph_adjacency_kernel    (:207-210): one wavefront per row in both, no (G, n, n) temporaries
//   pgraph_softmax_fwd_kernel  A_glob = softmax_row((1 - M) * sim)  (:211,214-215): local pairs keep logit 0 (not -inf:
//                              SURVEY fact 6), padding columns carry no mass; one wavefront per row, shuffle reductions
//   pgraph_softmax_bwd_kernel  its gradient with respect to sim
// The three contractions around them (A_local @ x, sam sam^T, A_glob @ mixed) stay batched MFMA GEMMs (htd_bgemm_nt).
#include "common.h"

namespace {

// (inter / max(union, eps)) > 0 with the arithmetic of bbox_overlaps (iou2d_calculator.py:148-150)
__device__ __forceinline__ bool overlaps(float4 a, float4 b)
{
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float inter = w * h;
    const float uni = fmaxf((a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y) - inter, 1e-6f);
    return inter / uni > 0.f;
}

// degrees: one wavefront per row, lanes across the columns (ballot + popcount); dinv = D^-1/2, 1 for padded rows
__global__ __launch_bounds__(256) void pgraph_degree_kernel(const float4 *__restrict__ boxes, const int64_t *__restrict__ counts,
                                                            float *__restrict__ dinv, int npad, int64_t rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int g = (int)(r / npad), i = (int)(r % npad);
    const int cnt = (int)counts[g];
    int deg = 0;
    if (i < cnt) {
        const float4 *bx = boxes + (size_t)g * npad;
        const float4 bi = bx[i];
        for (int j0 = 0; j0 < cnt; j0 += 64) {
            const int j = j0 + lane;
            const bool m = j < cnt && (j == i || overlaps(bi, bx[j]));
            deg += __popcll(__ballot(m));
        }
    }
    if (lane == 0) dinv[r] = 1.f / sqrtf((float)(deg > 0 ? deg : 1));
}

// A_local rows: one wavefront per row, A[i][j] = M_ij * dinv_i * dinv_j (zeros in the padding)
__global__ __launch_bounds__(256) void pgraph_adjacency_kernel(const float4 *__restrict__ boxes, const int64_t *__restrict__ counts,
                                                               const float *__restrict__ dinv, float *__restrict__ A, int npad,
                                                               int64_t rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int g = (int)(r / npad), i = (int)(r % npad);
    const int cnt = (int)counts[g];
    const float4 *bx = boxes + (size_t)g * npad;
    const float *dg = dinv + (size_t)g * npad;
    float *row = A + (size_t)r * npad;
    const bool vi = i < cnt;
    const float4 bi = vi ? bx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float di = dg[i];
    for (int j = lane; j < npad; j += 64) {
        const bool m = vi && j < cnt && (i == j || overlaps(bi, bx[j]));
        row[j] = m ? di * dg[j] : 0.f;
    }
}

constexpr int SM_MAX = 16;          // npad <= 64 * SM_MAX = 1024 columns per row

__global__ __launch_bounds__(256) void pgraph_softmax_fwd_kernel(const float *__restrict__ sim, const float *__restrict__ A_local,
                                                                 const int64_t *__restrict__ counts, float *__restrict__ A_glob,
                                                                 int npad, int64_t rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int g = (int)(r / npad), i = (int)(r % npad);
    const int cnt = (int)counts[g];
    const size_t off = (size_t)r * npad;
    if (i >= cnt) {                                     // padded row: zeros
        for (int j = lane; j < npad; j += 64) A_glob[off + j] = 0.f;
        return;
    }
    float v[SM_MAX];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < SM_MAX; ++t) {
        const int j = lane + 64 * t;
        v[t] = -INFINITY;
        if (j < npad && j < cnt) {
            const float m = A_local[off + j] > 0.f ? 1.f : 0.f;
            v[t] = (1.f - m) * sim[off + j];
            mx = fmaxf(mx, v[t]);
        }
    }
    mx = htd::wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < SM_MAX; ++t) {
        v[t] = (lane + 64 * t < cnt) ? expf(v[t] - mx) : 0.f;
        sum += v[t];
    }
    sum = htd::wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int t = 0; t < SM_MAX; ++t) {
        const int j = lane + 64 * t;
        if (j < npad) A_glob[off + j] = v[t] * inv;
    }
}

__global__ __launch_bounds__(256) void pgraph_softmax_bwd_kernel(const float *__restrict__ gA, const float *__restrict__ A_glob,
                                                                 const float *__restrict__ A_local, const int64_t *__restrict__ counts,
                                                                 float *__restrict__ gsim, int npad, int64_t rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int g = (int)(r / npad), i = (int)(r % npad);
    const int cnt = (int)counts[g];
    const size_t off = (size_t)r * npad;
    if (i >= cnt) {
        for (int j = lane; j < npad; j += 64) gsim[off + j] = 0.f;
        return;
    }
    float a[SM_MAX], ga[SM_MAX];
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < SM_MAX; ++t) {
        const int j = lane + 64 * t;
        a[t] = ga[t] = 0.f;
        if (j < cnt) {
            a[t] = A_glob[off + j];
            ga[t] = gA[off + j];
            dot += a[t] * ga[t];
        }
    }
    dot = htd::wave_sum(dot);
#pragma unroll
    for (int t = 0; t < SM_MAX; ++t) {
        const int j = lane + 64 * t;
        if (j < npad) {
            float o = 0.f;
            if (j < cnt) {
                const float m = A_local[off + j] > 0.f ? 1.f : 0.f;
                o = (1.f - m) * a[t] * (ga[t] - dot);
            }
            gsim[off + j] = o;
        }
    }
}

}  // namespace

// boxes [G][npad][4] (x1,y1,x2,y2; rows >= counts[g] ignored), counts [G] int64 on the device -> A_local [G][npad][npad].
// dinv: scratch of G * npad floats (receives D^-1/2).
extern "C" int htd_pgraph_adjacency(const float *boxes, const int64_t *counts, float *A_local, float *dinv, int G, int npad,
                                    void *stream)
{
    HTD_REQUIRE(G > 0 && npad > 0 && npad % 64 == 0 && npad <= 64 * SM_MAX, "pgraph_adjacency: bad sizes G=%d npad=%d", G, npad);
    HTD_REQUIRE(boxes && counts && A_local && dinv, "pgraph_adjacency: null pointer");
    const int64_t rows = (int64_t)G * npad;
    const unsigned blocks = (unsigned)htd::ceil_div(rows, 4);
    hipLaunchKernelGGL(pgraph_degree_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)boxes, counts, dinv, npad, rows);
    hipLaunchKernelGGL(pgraph_adjacency_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)boxes, counts, dinv,
                       A_local, npad, rows);
    return htd::check_launch("pgraph_adjacency");
}

// A_glob = rowwise softmax((1 - [A_local > 0]) * sim) over the counts[g] valid columns, rows >= counts[g] zero.
extern "C" int htd_pgraph_softmax_fwd(const float *sim, const float *A_local, const int64_t *counts, float *A_glob, int G, int npad,
                                      void *stream)
{
    HTD_REQUIRE(G > 0 && npad > 0 && npad % 64 == 0 && npad <= 64 * SM_MAX, "pgraph_softmax: bad sizes G=%d npad=%d", G, npad);
    HTD_REQUIRE(sim && A_local && counts && A_glob, "pgraph_softmax: null pointer");
    const int64_t rows = (int64_t)G * npad;
    hipLaunchKernelGGL(pgraph_softmax_fwd_kernel, dim3((unsigned)htd::ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, sim, A_local,
                       counts, A_glob, npad, rows);
    return htd::check_launch("pgraph_softmax_fwd");
}

extern "C" int htd_pgraph_softmax_bwd(const float *gA, const float *A_glob, const float *A_local, const int64_t *counts, float *gsim,
                                      int G, int npad, void *stream)
{
    HTD_REQUIRE(G > 0 && npad > 0 && npad % 64 == 0 && npad <= 64 * SM_MAX, "pgraph_softmax_bwd: bad sizes G=%d npad=%d", G, npad);
    HTD_REQUIRE(gA && A_glob && A_local && counts && gsim, "pgraph_softmax_bwd: null pointer");
    const int64_t rows = (int64_t)G * npad;
    hipLaunchKernelGGL(pgraph_softmax_bwd_kernel, dim3((unsigned)htd::ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, gA, A_glob,
                       A_local, counts, gsim, npad, rows);
    return htd::check_launch("pgraph_softmax_bwd");
}

// ---- group gather / scatter (pgraph.py): RoI rows -> padded (group, slot) rows and back -------------------------------
namespace {

// out[i][0..Fo) = valid[i] ? x[rows[i]][0..F) followed by zeros : zeros         (one wavefront per output row)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ x, const int64_t *__restrict__ rows,
                                                          const unsigned char *__restrict__ valid, float *__restrict__ out,
                                                          int64_t n_out, int F, int Fo)
{
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_out) return;
    const int lane = threadIdx.x & 63;
    const bool ok = valid[i] != 0;
    const float *src = x + (ok ? rows[i] : 0) * (int64_t)F;
    float *dst = out + i * (int64_t)Fo;
    for (int f = lane; f < Fo; f += 64) dst[f] = (ok && f < F) ? src[f] : 0.f;
}

// gx[rows[i]][0..F) = g[i][0..F) for the valid i (every RoI sits in at most one slot: plain stores); gx is zeroed first
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float *__restrict__ g, const int64_t *__restrict__ rows,
                                                           const unsigned char *__restrict__ valid, float *__restrict__ gx,
                                                           int64_t n_out, int F, int Fo)
{
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_out || !valid[i]) return;
    const int lane = threadIdx.x & 63;
    const float *src = g + i * (int64_t)Fo;
    float *dst = gx + rows[i] * (int64_t)F;
    for (int f = lane; f < F; f += 64) dst[f] = src[f];
}

// transposed forms: outT[g][f][r] = valid[g][r] ? x[rows[g][r]][f] : 0 through a 32 (slots) x 64 (features) LDS tile;
// TO_X: the adjoint, gx[rows[g][r]][f] = gT[g][f][r]
template <bool TO_X>
__global__ __launch_bounds__(256) void gather_rows_t_kernel(float *__restrict__ x, const int64_t *__restrict__ rows,
                                                            const unsigned char *__restrict__ valid, float *__restrict__ outT,
                                                            int npad, int F)
{
    __shared__ float tile[32][65];
    const int g = blockIdx.z, r0 = blockIdx.y * 32, f0 = blockIdx.x * 64, t = threadIdx.x;
    const int64_t slot0 = (int64_t)g * npad + r0;
    if (!TO_X) {
        for (int e = t; e < 32 * 64; e += 256) {
            const int r = e >> 6, f = e & 63;
            const bool ok = r0 + r < npad && f0 + f < F && valid[slot0 + r];
            tile[r][f] = ok ? x[rows[slot0 + r] * (int64_t)F + f0 + f] : 0.f;
        }
        __syncthreads();
        for (int e = t; e < 32 * 64; e += 256) {
            const int f = e >> 5, r = e & 31;
            if (r0 + r < npad && f0 + f < F) outT[((int64_t)g * F + f0 + f) * npad + r0 + r] = tile[r][f];
        }
    } else {
        for (int e = t; e < 32 * 64; e += 256) {
            const int f = e >> 5, r = e & 31;
            tile[r][f] = (r0 + r < npad && f0 + f < F) ? outT[((int64_t)g * F + f0 + f) * npad + r0 + r] : 0.f;
        }
        __syncthreads();
        for (int e = t; e < 32 * 64; e += 256) {
            const int r = e >> 6, f = e & 63;
            if (r0 + r < npad && f0 + f < F && valid[slot0 + r]) x[rows[slot0 + r] * (int64_t)F + f0 + f] = tile[r][f];
        }
    }
}

}  // namespace

// x [N][F] -> out [n_out][Fo] (Fo >= F, zero tail), out[i] = valid[i] ? x[rows[i]] : 0: `x[mask]` group gathers of
// HTDBBoxHead.forward (htd_bbox_head.py:198-206) for all groups at once.  transposed != 0: out is [G][F][npad] (n_out = G * npad,
// Fo == F), the K-major operand of the adjacency product.
extern "C" int htd_pgraph_gather(const float *x, const int64_t *rows, const unsigned char *valid, float *out, int64_t n_out,
                                 int F, int Fo, int G, int transposed, void *stream)
{
    HTD_REQUIRE(n_out >= 0 && F > 0 && Fo >= F && G > 0 && n_out % G == 0, "pgraph_gather: bad sizes");
    if (n_out == 0) return HTD_OK;
    HTD_REQUIRE(x && rows && valid && out, "pgraph_gather: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (transposed) {
        HTD_REQUIRE(Fo == F, "pgraph_gather: the transposed form has no padding columns");
        const int npad = (int)(n_out / G);
        const dim3 grid((unsigned)htd::ceil_div(F, 64), (unsigned)htd::ceil_div(npad, 32), (unsigned)G);
        hipLaunchKernelGGL(gather_rows_t_kernel<false>, grid, dim3(256), 0, s, const_cast<float *>(x), rows, valid, out, npad, F);
    } else
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)htd::ceil_div(n_out, 4)), dim3(256), 0, s, x, rows, valid, out, n_out,
                           F, Fo);
    return htd::check_launch("pgraph_gather");
}

// The adjoint: gx [N][F] = 0, then gx[rows[i]] = g[i][0..F) for the valid i (each RoI occupies at most one slot).
extern "C" int htd_pgraph_scatter(const float *g, const int64_t *rows, const unsigned char *valid, float *gx, int64_t n_out,
                                  int64_t N, int F, int Fo, int G, int transposed, void *stream)
{
    HTD_REQUIRE(n_out >= 0 && N >= 0 && F > 0 && Fo >= F && G > 0 && n_out % G == 0, "pgraph_scatter: bad sizes");
    if (N == 0) return HTD_OK;
    HTD_REQUIRE(gx, "pgraph_scatter: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(gx, 0, (size_t)N * F * sizeof(float), s) != hipSuccess) {
        htd::set_error("pgraph_scatter: memset failed");
        return HTD_ERR_LAUNCH;
    }
    if (n_out == 0) return HTD_OK;
    HTD_REQUIRE(g && rows && valid, "pgraph_scatter: null pointer");
    if (transposed) {
        HTD_REQUIRE(Fo == F, "pgraph_scatter: the transposed form has no padding columns");
        const int npad = (int)(n_out / G);
        const dim3 grid((unsigned)htd::ceil_div(F, 64), (unsigned)htd::ceil_div(npad, 32), (unsigned)G);
        hipLaunchKernelGGL(gather_rows_t_kernel<true>, grid, dim3(256), 0, s, gx, rows, valid, const_cast<float *>(g), npad, F);
    } else
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)htd::ceil_div(n_out, 4)), dim3(256), 0, s, g, rows, valid, gx, n_out,
                           F, Fo);
    return htd::check_launch("pgraph_scatter");
}
