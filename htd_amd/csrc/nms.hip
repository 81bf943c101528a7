// Greedy hard NMS on score-sorted boxes, batched over independent segments
// ((image, pyramid level) for the RPN, (image, class) for the final detections), entirely
// on the device: no host round trip between the pairwise pass and the greedy pass.
//
// Pass A (nms_mask_kernel): one wave per 64x64 tile of the upper triangle; lane t owns row
//   t of the tile and emits a 64-bit word = which of the tile's 64 column boxes it would
//   suppress (wave64 <-> one uint64 word, no 32-bit splitting as in warp-shaped code).
// Pass B (nms_reduce_kernel): one wave per segment walks its rows in chunks of 64: the
//   chunk's diagonal tile is resolved with 64 scalar steps on register data, then every lane
//   ORs the surviving rows' words into the running "removed" words of the column tiles it
//   owns (lane l owns tiles l, l+64, ...), reading the mask exactly once, coalesced.
//
// The suppression test is the CPU path's arithmetic: inter / (areaA + areaB - inter) > thr
// with an IEEE fp32 division, compiled with -ffp-contract=off so `union` is not fused.
#include "common.h"

namespace {

constexpr int KMAX = 4;  // column tiles per lane in pass B -> at most 64*64*KMAX boxes / segment

__device__ __forceinline__ bool suppress(const float4 a, float area_a, const float4 b, float area_b,
                                         float thr, float off)
{
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float w = fmaxf(0.f, xx2 - xx1 + off), h = fmaxf(0.f, yy2 - yy1 + off);
    const float inter = w * h;
    const float ovr = inter / (area_a + area_b - inter);
    return ovr > thr;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4 *__restrict__ boxes,
                                                      const int64_t *__restrict__ seg_offsets,
                                                      unsigned long long *__restrict__ mask, int ncb,
                                                      float thr, float off)
{
    const int cb = blockIdx.x, rb = blockIdx.y, seg = blockIdx.z;
    if (cb < rb) return;
    const int64_t begin = seg_offsets[seg];
    const int64_t n = seg_offsets[seg + 1] - begin;
    if ((int64_t)rb * 64 >= n || (int64_t)cb * 64 >= n) return;
    const int t = threadIdx.x;
    __shared__ float4 cbox[64];
    __shared__ float carea[64];
    const int64_t cj = (int64_t)cb * 64 + t;
    if (cj < n) {
        const float4 b = boxes[begin + cj];
        cbox[t] = b;
        carea[t] = (b.z - b.x + off) * (b.w - b.y + off);
    }
    __syncthreads();
    const int64_t ri = (int64_t)rb * 64 + t;
    if (ri >= n) return;
    const float4 a = boxes[begin + ri];
    const float area_a = (a.z - a.x + off) * (a.w - a.y + off);
    const int ncols = (int)min((int64_t)64, n - (int64_t)cb * 64);
    unsigned long long word = 0ull;
    const int jstart = (cb == rb) ? t + 1 : 0;
    for (int j = jstart; j < ncols; ++j)
        if (suppress(a, area_a, cbox[j], carea[j], thr, off)) word |= 1ull << j;
    mask[(begin + ri) * ncb + cb] = word;
}

__device__ __forceinline__ unsigned long long bcast64(unsigned long long v, int src)
{
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, src);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void nms_reduce_kernel(const int64_t *__restrict__ seg_offsets,
                                                        const unsigned long long *__restrict__ mask,
                                                        uint8_t *__restrict__ keep, int ncb)
{
    const int seg = blockIdx.x, lane = threadIdx.x;
    const int64_t begin = seg_offsets[seg];
    const int64_t n = seg_offsets[seg + 1] - begin;
    if (n <= 0) return;
    const int nchunks = (int)((n + 63) / 64);
    unsigned long long remv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) remv[k] = 0ull;

    for (int q = 0; q < nchunks; ++q) {
        const int64_t row0 = begin + (int64_t)q * 64;
        const int rows = (int)min((int64_t)64, n - (int64_t)q * 64);
        // removed-word of this chunk lives in lane q%64, slot q/64
        unsigned long long cur = 0ull;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if ((q >> 6) == k) cur = bcast64(remv[k], q & 63);
        if (rows < 64) cur |= ~0ull << rows;  // rows past the end are never kept
        // upper-triangular mask: no row before this chunk can change after this point
        const unsigned long long dword = lane < rows ? mask[(row0 + lane) * ncb + q] : 0ull;
        // the words this lane will fold (column tile `lane`, all rows of the chunk) do not depend on which rows survive:
        // their 64 loads are issued now and land while the diagonal tile is resolved -- the walk over the chunks is one
        // dependent chain per segment, so every exposed load latency is paid n / 64 times
        const bool own0 = lane > q && lane < ncb && (int64_t)lane * 64 < n;
        unsigned long long pre[64];
        if (own0) {
            const unsigned long long *col = mask + row0 * ncb + lane;
#pragma unroll
            for (int t = 0; t < 64; ++t) pre[t] = col[(int64_t)min(t, rows - 1) * ncb];
        }
        for (int t = 0; t < rows; ++t) {
            const unsigned long long dt = bcast64(dword, t);
            if (!((cur >> t) & 1ull)) cur |= dt;
        }
        const unsigned long long kept = ~cur;      // bits of the rows past the end are 0
        if (lane < rows) keep[row0 + lane] = (uint8_t)((kept >> lane) & 1ull);
        // fold the survivors' rows into the later column tiles this lane owns
        if (own0) {
            unsigned long long acc = remv[0];
#pragma unroll
            for (int t = 0; t < 64; ++t) acc |= ((kept >> t) & 1ull) ? pre[t] : 0ull;
            remv[0] = acc;
        }
#pragma unroll
        for (int k = 1; k < KMAX; ++k) {
            const int cb = lane + 64 * k;
            if (cb > q && cb < ncb && (int64_t)cb * 64 < n) {
                unsigned long long acc = remv[k];
                const unsigned long long *col = mask + row0 * ncb + cb;
#pragma unroll 8
                for (int t = 0; t < rows; ++t) {
                    const unsigned long long m = col[(int64_t)t * ncb];
                    acc |= ((kept >> t) & 1ull) ? m : 0ull;
                }
                remv[k] = acc;
            }
        }
    }
}

__global__ void zero_words(unsigned long long *p, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = 0ull;
}

int64_t ncb_for(int64_t max_seg) { return (max_seg + 63) / 64; }

}  // namespace

extern "C" int64_t htd_nms_workspace_bytes(int64_t n_total)
{
    // worst case: one segment holding everything, capped by the per-segment limit
    const int64_t cap = 64ll * 64 * KMAX;
    const int64_t ms = n_total < cap ? n_total : cap;
    return n_total * ncb_for(ms) * 8 + 64;
}

extern "C" int htd_nms_sorted_batched(const float *boxes, const int64_t *seg_offsets, int segments,
                                      int64_t n_total, int64_t max_seg, uint8_t *keep_mask, float iou_thr,
                                      int offset, void *workspace, void *stream)
{
    HTD_REQUIRE(segments >= 0 && n_total >= 0 && max_seg >= 0, "nms: negative size");
    HTD_REQUIRE(max_seg <= 64ll * 64 * KMAX, "nms: segment of %lld boxes exceeds the %d-box limit",
                (long long)max_seg, 64 * 64 * KMAX);
    HTD_REQUIRE(offset == 0 || offset == 1, "nms: offset must be 0 or 1");
    if (segments == 0 || n_total == 0 || max_seg == 0) return HTD_OK;
    HTD_REQUIRE(boxes && seg_offsets && keep_mask && workspace, "nms: null pointer");
    HTD_REQUIRE(((uintptr_t)boxes & 15) == 0, "nms: boxes must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int ncb = (int)ncb_for(max_seg);
    auto *mask = (unsigned long long *)workspace;
    // tiles below the diagonal / past a short segment's end are read by pass B only where
    // written, but zero the buffer so partially filled rows are well defined
    hipLaunchKernelGGL(zero_words, dim3(1024), dim3(256), 0, s, mask, n_total * ncb);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(ncb, ncb, segments), dim3(64), 0, s, (const float4 *)boxes,
                       seg_offsets, mask, ncb, iou_thr, (float)offset);
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(segments), dim3(64), 0, s, seg_offsets, mask, keep_mask, ncb);
    return htd::check_launch("nms");
}

namespace {
__global__ void write_two(int64_t *p, int64_t n) { p[0] = 0; p[1] = n; }
}

extern "C" int htd_nms_sorted(const float *boxes, uint8_t *keep_mask, int64_t n, float iou_thr, int offset,
                              void *workspace, void *stream)
{
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(workspace, "nms: null workspace");
    // the segment table of the single-problem form lives in the last 64 bytes of the workspace
    const int64_t ncb = ncb_for(n);
    int64_t *seg = (int64_t *)((char *)workspace + n * ncb * 8);
    seg = (int64_t *)(((uintptr_t)seg + 15) & ~(uintptr_t)15);
    hipLaunchKernelGGL(write_two, dim3(1), dim3(1), 0, (hipStream_t)stream, seg, n);
    return htd_nms_sorted_batched(boxes, seg, 1, n, n, keep_mask, iou_thr, offset, workspace, stream);
}


// ------------------------------------------------------------------------------------------------------------
// Soft-NMS (mmcv.ops.soft_nms; R101 test configs: configs/htd/htd_resnet101_2x.py:298 `type='soft_nms'`,
// linear decay, min_score 0.05).  mmcv runs it sequentially on the CPU after a device->host copy
// (build/lib/mmdet/ops/nms/nms_wrapper.py:62-116).  Here every (image, class) segment is one workgroup: each
// round picks the best remaining box (block-wide arg-max), then decays / discards the rest in parallel.  Rounds are
// inherently sequential, segments are independent.  Arithmetic as the CPU code: ovr = inter / (a_i + a_j - inter),
// linear: w = 1 - ovr if ovr >= thr; naive: w = 0; gaussian: w = exp(-ovr^2 / sigma); box dropped when its score
// falls below min_score.  Equal scores are resolved towards the lower row (the CPU code's swap history decides).
namespace {

constexpr int SN_THREADS = 256;

__global__ __launch_bounds__(SN_THREADS) void soft_nms_kernel(const float4 *__restrict__ boxes, float *__restrict__ scores,
                                                              const int64_t *__restrict__ seg_offsets,
                                                              int *__restrict__ rank, float thr, float sigma,
                                                              float min_score, int method, float off)
{
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t begin = seg_offsets[seg];
    const int n = (int)(seg_offsets[seg + 1] - begin);
    const float4 *bx = boxes + begin;
    float *sc = scores + begin;
    int *rk = rank + begin;                      // -1 active, -2 discarded, >= 0 selection round
    for (int i = tid; i < n; i += SN_THREADS) rk[i] = -1;
    __shared__ float red_s[SN_THREADS / 64];
    __shared__ int red_i[SN_THREADS / 64];
    __shared__ int chosen;
    __syncthreads();
    for (int round = 0; round < n; ++round) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += SN_THREADS)
            if (rk[i] == -1) {
                const float v = sc[i];
                if (v > best || (v == best && i < bi)) { best = v; bi = i; }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { red_s[wave] = best; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < SN_THREADS / 64; ++w)
                if (red_s[w] > best || (red_s[w] == best && red_i[w] < bi)) { best = red_s[w]; bi = red_i[w]; }
            chosen = bi;
            if (bi != 0x7fffffff) rk[bi] = round;
        }
        __syncthreads();
        const int c = chosen;
        if (c == 0x7fffffff) break;
        const float4 a = bx[c];
        const float area_a = (a.z - a.x + off) * (a.w - a.y + off);
        for (int i = tid; i < n; i += SN_THREADS)
            if (rk[i] == -1) {
                const float4 b = bx[i];
                const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
                const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
                const float w = fmaxf(0.f, xx2 - xx1 + off), h = fmaxf(0.f, yy2 - yy1 + off);
                const float inter = w * h;
                const float area_b = (b.z - b.x + off) * (b.w - b.y + off);
                const float ovr = inter / (area_a + area_b - inter);
                float weight = 1.f;
                if (method == 0) { if (ovr >= thr) weight = 0.f; }
                else if (method == 1) { if (ovr >= thr) weight = 1.f - ovr; }
                else weight = expf(-(ovr * ovr) / sigma);
                const float v = sc[i] * weight;
                sc[i] = v;
                if (v < min_score) rk[i] = -2;
            }
        __syncthreads();
    }
}

}  // namespace

extern "C" int htd_soft_nms_segments(const float *boxes, float *scores, const int64_t *seg_offsets, int segments,
                                     int64_t n_total, int *rank, float iou_thr, float sigma, float min_score,
                                     int method, int offset, void *stream)
{
    HTD_REQUIRE(segments >= 0 && n_total >= 0, "soft_nms: negative size");
    HTD_REQUIRE(method >= 0 && method <= 2, "soft_nms: method must be 0 (naive), 1 (linear) or 2 (gaussian)");
    HTD_REQUIRE(offset == 0 || offset == 1, "soft_nms: offset must be 0 or 1");
    if (segments == 0 || n_total == 0) return HTD_OK;
    HTD_REQUIRE(boxes && scores && seg_offsets && rank, "soft_nms: null pointer");
    hipLaunchKernelGGL(soft_nms_kernel, dim3((unsigned)segments), dim3(SN_THREADS), 0, (hipStream_t)stream,
                       (const float4 *)boxes, scores, seg_offsets, rank, iou_thr, sigma, min_score, method, (float)offset);
    return htd::check_launch("soft_nms");
}
