// PGraph adjacency kernels (HTDBBoxHead.forward, roi_heads/bbox_heads/htd_bbox_head.py:198-219), batched over the
// (image, pyramid level) groups of a call: group g holds counts[g] RoIs in rows 0..counts[g]-1 of a [G][npad] padding.
//
//   pgraph_adjacency_kernel    boxes -> A_local = D^-1/2 M D^-1/2,  M = (IoU with unit diagonal) > 0,  D = rowsum(M)
//                              (:207-210): IoU, mask, degree and normalisation in ONE pass, no (G, n, n) temporaries
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

// grid (npad / 64, G), 256 threads: a workgroup owns 64 rows of one group.  Degrees of ALL rows of the group are
// recomputed by every workgroup of the group into LDS (n^2 / 256 box tests per thread; n <= ~1000: a few microseconds)
// so that the normalisation needs no second launch.
__global__ __launch_bounds__(256) void pgraph_adjacency_kernel(const float4 *__restrict__ boxes, const int64_t *__restrict__ counts,
                                                               float *__restrict__ A, int npad)
{
    extern __shared__ float dinv[];                  // [npad]
    float4 *sbox = reinterpret_cast<float4 *>(dinv + npad);       // [npad]
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cnt = (int)counts[g];
    const float4 *bx = boxes + (size_t)g * npad;
    for (int j = tid; j < npad; j += 256) sbox[j] = j < cnt ? bx[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int j = tid; j < npad; j += 256) {
        int deg = 0;
        if (j < cnt) {
            const float4 bj = sbox[j];
            for (int k = 0; k < cnt; ++k) deg += (k == j || overlaps(bj, sbox[k])) ? 1 : 0;
        }
        dinv[j] = 1.f / sqrtf((float)(deg > 0 ? deg : 1));                 // padded rows: 1 (never used)
    }
    __syncthreads();
    const int i0 = blockIdx.x * 64 + wave * 16;
    for (int i = i0; i < i0 + 16 && i < npad; ++i) {
        float *row = A + ((size_t)g * npad + i) * npad;
        const bool vi = i < cnt;
        const float4 bi = sbox[i];
        const float di = dinv[i];
        for (int j = lane; j < npad; j += 64) {
            const bool m = vi && j < cnt && (i == j || overlaps(bi, sbox[j]));
            row[j] = m ? di * dinv[j] : 0.f;
        }
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
extern "C" int htd_pgraph_adjacency(const float *boxes, const int64_t *counts, float *A_local, int G, int npad, void *stream)
{
    HTD_REQUIRE(G > 0 && npad > 0 && npad % 64 == 0 && npad <= 64 * SM_MAX, "pgraph_adjacency: bad sizes G=%d npad=%d", G, npad);
    HTD_REQUIRE(boxes && counts && A_local, "pgraph_adjacency: null pointer");
    HTD_REQUIRE(G <= 65535, "pgraph_adjacency: too many groups");
    hipLaunchKernelGGL(pgraph_adjacency_kernel, dim3((unsigned)(npad / 64), (unsigned)G), dim3(256), (size_t)npad * 20, (hipStream_t)stream,
                       (const float4 *)boxes, counts, A_local, npad);
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
