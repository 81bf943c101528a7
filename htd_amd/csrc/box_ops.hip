// Box bookkeeping of the RPN / RoI heads on the device (gfx950): MaxIoUAssigner for a whole batch.
//
// Compiled with -ffp-contract=off: the IoU must be the exact fp32 expression of the reference
// (bbox_overlaps, iou2d_calculator.py:43-158: overlap / max(area_a + area_b - overlap, eps)) so that the
// threshold comparisons and the "all ties" low-quality matching select the same anchors bit for bit.
#include "common.h"

namespace {

struct AssignParams {
    const float *boxes;          // [B or 1][A][4]
    int64_t box_bstride;         // 0: anchors shared by all images
    const uint8_t *box_valid;    // [B][A]
    const float *gts;            // [B][K][4] zero padded
    const uint8_t *gt_valid;     // [B][K]
    int A, K;
    float pos_thr, neg_thr, min_pos;
    int low_quality;
    int64_t *assigned;           // [B][A]
    float *max_ov;               // [B][A]
    unsigned *gt_max;            // [B][K] float bits (IoU >= 0), zero-initialised by pass 1's caller
};

__device__ __forceinline__ float iou_of(float4 g, float ga, float4 b, float ba)
{
    const float w = fmaxf(fminf(g.z, b.z) - fmaxf(g.x, b.x), 0.f);
    const float h = fmaxf(fminf(g.w, b.w) - fmaxf(g.y, b.y), 0.f);
    const float overlap = w * h;
    const float uni = fmaxf(ga + ba - overlap, 1e-6f);
    return overlap / uni;
}

constexpr int GT_CHUNK = 128;

// pass 1: per box the best gt (first maximum), thresholds; per gt the best IoU over the valid boxes (wave max -> one
// integer atomicMax per wave: IoUs are non-negative, so their bit patterns order like the floats).
__global__ __launch_bounds__(256) void assign_pass1_kernel(AssignParams p)
{
    __shared__ float4 sg[GT_CHUNK];
    __shared__ float sga[GT_CHUNK];
    __shared__ int sok[GT_CHUNK];
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool in = a < p.A;
    const bool bvalid = in && p.box_valid[(int64_t)b * p.A + a] != 0;
    float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) box = *reinterpret_cast<const float4 *>(p.boxes + (int64_t)b * p.box_bstride + (int64_t)a * 4);
    const float ba = (box.z - box.x) * (box.w - box.y);
    float best = -1.f;
    int arg = 0;
    bool any_gt = false;
    for (int k0 = 0; k0 < p.K; k0 += GT_CHUNK) {
        const int kn = min(GT_CHUNK, p.K - k0);
        __syncthreads();
        if (threadIdx.x < kn) {
            const float4 g = *reinterpret_cast<const float4 *>(p.gts + ((int64_t)b * p.K + k0 + threadIdx.x) * 4);
            sg[threadIdx.x] = g;
            sga[threadIdx.x] = (g.z - g.x) * (g.w - g.y);
            sok[threadIdx.x] = p.gt_valid[(int64_t)b * p.K + k0 + threadIdx.x];
        }
        __syncthreads();
        for (int k = 0; k < kn; ++k) {
            if (!sok[k]) continue;                       // uniform
            any_gt = true;
            const float v = bvalid ? iou_of(sg[k], sga[k], box, ba) : -1.f;
            if (v > best) { best = v; arg = k0 + k; }
            if (p.low_quality) {
                const float m = htd::wave_max(v);
                if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(p.gt_max + (int64_t)b * p.K + k0 + k, __float_as_uint(m));
            }
        }
    }
    if (!in) return;
    int64_t out = -1;
    if (best >= 0.f && best < p.neg_thr) out = 0;
    if (best >= p.pos_thr) out = arg + 1;
    if (!any_gt) out = 0;
    if (!bvalid) out = -1;
    p.assigned[(int64_t)b * p.A + a] = out;
    p.max_ov[(int64_t)b * p.A + a] = (bvalid && any_gt) ? fmaxf(best, 0.f) : 0.f;
}

// pass 2 (match_low_quality, gt_max_assign_all: max_iou_assigner.py:187-199): every box whose IoU with gt k equals
// that gt's best IoU (>= min_pos_iou) is assigned to k; the last such gt wins.
__global__ __launch_bounds__(256) void assign_pass2_kernel(AssignParams p)
{
    __shared__ float4 sg[GT_CHUNK];
    __shared__ float sga[GT_CHUNK];
    __shared__ float smax[GT_CHUNK];                     // < 0: gt does not take part
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const bool in = a < p.A;
    const bool bvalid = in && p.box_valid[(int64_t)b * p.A + a] != 0;
    float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) box = *reinterpret_cast<const float4 *>(p.boxes + (int64_t)b * p.box_bstride + (int64_t)a * 4);
    const float ba = (box.z - box.x) * (box.w - box.y);
    int low = 0;
    for (int k0 = 0; k0 < p.K; k0 += GT_CHUNK) {
        const int kn = min(GT_CHUNK, p.K - k0);
        __syncthreads();
        if (threadIdx.x < kn) {
            const int64_t gi = (int64_t)b * p.K + k0 + threadIdx.x;
            const float4 g = *reinterpret_cast<const float4 *>(p.gts + gi * 4);
            sg[threadIdx.x] = g;
            sga[threadIdx.x] = (g.z - g.x) * (g.w - g.y);
            const float m = __uint_as_float(p.gt_max[gi]);
            smax[threadIdx.x] = (p.gt_valid[gi] && m >= p.min_pos) ? m : -1.f;
        }
        __syncthreads();
        if (bvalid)
            for (int k = 0; k < kn; ++k)
                if (smax[k] >= 0.f && iou_of(sg[k], sga[k], box, ba) == smax[k]) low = k0 + k + 1;
    }
    if (bvalid && low > 0) p.assigned[(int64_t)b * p.A + a] = low;              // invalid boxes stay -1
}

}  // namespace

// MaxIoUAssigner.assign_wrt_overlaps (core/bbox/assigners/max_iou_assigner.py:124-212) with gt_max_assign_all for B
// images in one or two launches.  boxes [B][A][4] (box_shared = 0) or [A][4] shared by all images (box_shared = 1);
// box_valid [B][A]; gts [B][K][4] zero-padded with gt_valid [B][K].  assigned [B][A] int64: -1 ignore / invalid box,
// 0 negative, k+1 matched to gt k; max_overlaps [B][A].  workspace: B*K*4 bytes when match_low_quality.
extern "C" int htd_max_iou_assign(const float *boxes, int box_shared, const uint8_t *box_valid, const float *gts,
                                  const uint8_t *gt_valid, int B, int A, int K, float pos_iou_thr, float neg_iou_thr,
                                  float min_pos_iou, int match_low_quality, int64_t *assigned, float *max_overlaps,
                                  void *workspace, void *stream)
{
    HTD_REQUIRE(B >= 0 && A >= 0 && K >= 1, "max_iou_assign: bad sizes B=%d A=%d K=%d", B, A, K);
    if (B == 0 || A == 0) return HTD_OK;
    HTD_REQUIRE(boxes && box_valid && gts && gt_valid && assigned && max_overlaps, "max_iou_assign: null pointer");
    HTD_REQUIRE(!match_low_quality || workspace, "max_iou_assign: match_low_quality needs a B*K*4 byte workspace");
    AssignParams p{};
    p.boxes = boxes; p.box_bstride = box_shared ? 0 : (int64_t)A * 4; p.box_valid = box_valid;
    p.gts = gts; p.gt_valid = gt_valid; p.A = A; p.K = K;
    p.pos_thr = pos_iou_thr; p.neg_thr = neg_iou_thr; p.min_pos = min_pos_iou; p.low_quality = match_low_quality;
    p.assigned = assigned; p.max_ov = max_overlaps; p.gt_max = (unsigned *)workspace;
    hipStream_t s = (hipStream_t)stream;
    if (match_low_quality && hipMemsetAsync(workspace, 0, (size_t)B * K * 4, s) != hipSuccess) {
        htd::set_error("max_iou_assign: memset failed");
        return HTD_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)htd::ceil_div(A, 256), (unsigned)B);
    hipLaunchKernelGGL(assign_pass1_kernel, grid, dim3(256), 0, s, p);
    if (match_low_quality) hipLaunchKernelGGL(assign_pass2_kernel, grid, dim3(256), 0, s, p);
    return htd::check_launch("max_iou_assign");
}
