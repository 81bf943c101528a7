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
    float *gt_max;               // [B][K] best IoU of each gt over the valid boxes (-1: none)
    float *block_max;            // [B][blocks][K] per-workgroup maxima of pass 1
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

// pass 1: per box the best gt (first maximum), thresholds; per gt this workgroup's best IoU over its valid boxes.
__global__ __launch_bounds__(256) void assign_pass1_kernel(AssignParams p)
{
    __shared__ float4 sg[GT_CHUNK];
    __shared__ float sga[GT_CHUNK];
    __shared__ int sok[GT_CHUNK];
    __shared__ float swave[4][GT_CHUNK];
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
                if ((threadIdx.x & 63) == 0) swave[threadIdx.x >> 6][k] = m;
            }
        }
        if (p.low_quality) {                 // this block's best IoU per gt of the chunk (no atomics: B*K hot addresses)
            __syncthreads();
            if (threadIdx.x < kn) {
                const float m = sok[threadIdx.x]
                    ? fmaxf(fmaxf(swave[0][threadIdx.x], swave[1][threadIdx.x]), fmaxf(swave[2][threadIdx.x], swave[3][threadIdx.x]))
                    : -1.f;
                p.block_max[((int64_t)b * gridDim.x + blockIdx.x) * p.K + k0 + threadIdx.x] = m;
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

// best IoU of every gt over all workgroups of its image
__global__ __launch_bounds__(256) void assign_gtmax_kernel(AssignParams p, int blocks)
{
    __shared__ float red[256];
    const int64_t bk = blockIdx.x;                       // b * K + k
    const int64_t b = bk / p.K, k = bk - b * p.K;
    float m = -1.f;
    for (int i = threadIdx.x; i < blocks; i += 256) m = fmaxf(m, p.block_max[(b * blocks + i) * p.K + k]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) p.gt_max[bk] = red[0];
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
            const float m = p.gt_max[gi];
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
// 0 negative, k+1 matched to gt k; max_overlaps [B][A].  workspace: htd_max_iou_assign_workspace_bytes when
// match_low_quality.
extern "C" int64_t htd_max_iou_assign_workspace_bytes(int B, int A, int K)
{
    return ((int64_t)B * K + (int64_t)B * htd::ceil_div(A, 256) * K) * 4;
}

extern "C" int htd_max_iou_assign(const float *boxes, int box_shared, const uint8_t *box_valid, const float *gts,
                                  const uint8_t *gt_valid, int B, int A, int K, float pos_iou_thr, float neg_iou_thr,
                                  float min_pos_iou, int match_low_quality, int64_t *assigned, float *max_overlaps,
                                  void *workspace, void *stream)
{
    HTD_REQUIRE(B >= 0 && A >= 0 && K >= 1, "max_iou_assign: bad sizes B=%d A=%d K=%d", B, A, K);
    if (B == 0 || A == 0) return HTD_OK;
    HTD_REQUIRE(boxes && box_valid && gts && gt_valid && assigned && max_overlaps, "max_iou_assign: null pointer");
    HTD_REQUIRE(!match_low_quality || workspace, "max_iou_assign: match_low_quality needs a workspace");
    AssignParams p{};
    p.boxes = boxes; p.box_bstride = box_shared ? 0 : (int64_t)A * 4; p.box_valid = box_valid;
    p.gts = gts; p.gt_valid = gt_valid; p.A = A; p.K = K;
    p.pos_thr = pos_iou_thr; p.neg_thr = neg_iou_thr; p.min_pos = min_pos_iou; p.low_quality = match_low_quality;
    p.assigned = assigned; p.max_ov = max_overlaps;
    p.gt_max = (float *)workspace;
    p.block_max = p.gt_max + (int64_t)B * K;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)htd::ceil_div(A, 256), (unsigned)B);
    hipLaunchKernelGGL(assign_pass1_kernel, grid, dim3(256), 0, s, p);
    if (match_low_quality) {
        hipLaunchKernelGGL(assign_gtmax_kernel, dim3((unsigned)(B * K)), dim3(256), 0, s, p, (int)grid.x);
        hipLaunchKernelGGL(assign_pass2_kernel, grid, dim3(256), 0, s, p);
    }
    return htd::check_launch("max_iou_assign");
}

namespace {

struct Vec4 { float v[4]; };

// bbox_head.get_targets on fixed sample slots: label / weights / encoded regression target of every slot.
__global__ __launch_bounds__(256) void roi_targets_kernel(const float *__restrict__ boxes, const float *__restrict__ gt,
                                                          const int64_t *__restrict__ gt_labels,
                                                          const uint8_t *__restrict__ is_pos,
                                                          const uint8_t *__restrict__ valid, int64_t n, int num_classes,
                                                          Vec4 means, Vec4 stds, int64_t *__restrict__ labels,
                                                          float *__restrict__ label_w, float *__restrict__ targets,
                                                          float *__restrict__ bbox_w)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool pos = is_pos[i] != 0;
    labels[i] = pos ? gt_labels[i] : (int64_t)num_classes;
    label_w[i] = valid[i] ? 1.f : 0.f;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos) {      // bbox2delta (delta_xywh_bbox_coder.py:78-120)
        const float4 p = *reinterpret_cast<const float4 *>(boxes + i * 4);
        const float4 g = *reinterpret_cast<const float4 *>(gt + i * 4);
        const float px = (p.x + p.z) * 0.5f, py = (p.y + p.w) * 0.5f, pw = p.z - p.x, ph = p.w - p.y;
        const float gx = (g.x + g.z) * 0.5f, gy = (g.y + g.w) * 0.5f, gw = g.z - g.x, gh = g.w - g.y;
        t.x = ((gx - px) / pw - means.v[0]) / stds.v[0];
        t.y = ((gy - py) / ph - means.v[1]) / stds.v[1];
        t.z = (logf(gw / pw) - means.v[2]) / stds.v[2];
        t.w = (logf(gh / ph) - means.v[3]) / stds.v[3];
    }
    *reinterpret_cast<float4 *>(targets + i * 4) = t;
    const float w = pos ? 1.f : 0.f;
    *reinterpret_cast<float4 *>(bbox_w + i * 4) = make_float4(w, w, w, w);
}

// delta2bbox (delta_xywh_bbox_coder.py:123-204) for 4-column deltas, clipped to the image of each row
// (row / rows_per_img), rows with keep == 0 zeroed.
__global__ __launch_bounds__(256) void decode_clip_kernel(const float *__restrict__ rois, const float *__restrict__ deltas,
                                                          const float *__restrict__ lim_wh,
                                                          const uint8_t *__restrict__ keep, int64_t n,
                                                          int64_t rows_per_img, Vec4 means, Vec4 stds, float max_ratio,
                                                          float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!keep || keep[i]) {
        const float4 r = *reinterpret_cast<const float4 *>(rois + i * 4);
        const float4 d0 = *reinterpret_cast<const float4 *>(deltas + i * 4);
        const float dx = d0.x * stds.v[0] + means.v[0], dy = d0.y * stds.v[1] + means.v[1];
        float dw = d0.z * stds.v[2] + means.v[2], dh = d0.w * stds.v[3] + means.v[3];
        dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
        dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
        const float px = (r.x + r.z) * 0.5f, py = (r.y + r.w) * 0.5f, pw = r.z - r.x, ph = r.w - r.y;
        const float gw = pw * expf(dw), gh = ph * expf(dh);
        const float gx = px + pw * dx, gy = py + ph * dy;
        o = make_float4(gx - gw * 0.5f, gy - gh * 0.5f, gx + gw * 0.5f, gy + gh * 0.5f);
        if (lim_wh) {
            const int64_t b = i / rows_per_img;
            const float W = lim_wh[2 * b], H = lim_wh[2 * b + 1];
            o.x = fminf(fmaxf(o.x, 0.f), W); o.y = fminf(fmaxf(o.y, 0.f), H);
            o.z = fminf(fmaxf(o.z, 0.f), W); o.w = fminf(fmaxf(o.w, 0.f), H);
        }
    }
    *reinterpret_cast<float4 *>(out + i * 4) = o;
}

}  // namespace

extern "C" int htd_roi_targets(const float *boxes, const float *gt_boxes, const int64_t *gt_labels, const uint8_t *is_pos,
                               const uint8_t *valid, int64_t n, int num_classes, const float *means4,
                               const float *stds4, int64_t *labels, float *label_weights, float *bbox_targets,
                               float *bbox_weights, void *stream)
{
    HTD_REQUIRE(n >= 0 && num_classes > 0, "roi_targets: bad sizes");
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(boxes && gt_boxes && gt_labels && is_pos && valid && means4 && stds4 && labels && label_weights &&
                    bbox_targets && bbox_weights, "roi_targets: null pointer");
    Vec4 m, sd;
    for (int k = 0; k < 4; ++k) { m.v[k] = means4[k]; sd.v[k] = stds4[k]; }
    hipLaunchKernelGGL(roi_targets_kernel, dim3((unsigned)htd::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       boxes, gt_boxes, gt_labels, is_pos, valid, n, num_classes, m, sd, labels, label_weights,
                       bbox_targets, bbox_weights);
    return htd::check_launch("roi_targets");
}

extern "C" int htd_delta2bbox_clip(const float *rois, const float *deltas, const float *lim_wh, const uint8_t *keep,
                                   int64_t n, int64_t rows_per_img, const float *means4, const float *stds4,
                                   float wh_ratio_clip, float *out, void *stream)
{
    HTD_REQUIRE(n >= 0 && rows_per_img > 0 && wh_ratio_clip > 0.f, "delta2bbox_clip: bad sizes");
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(rois && deltas && means4 && stds4 && out, "delta2bbox_clip: null pointer");
    Vec4 m, sd;
    for (int k = 0; k < 4; ++k) { m.v[k] = means4[k]; sd.v[k] = stds4[k]; }
    const float max_ratio = fabsf(logf(wh_ratio_clip));
    hipLaunchKernelGGL(decode_clip_kernel, dim3((unsigned)htd::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       rois, deltas, lim_wh, keep, n, rows_per_img, m, sd, max_ratio, out);
    return htd::check_launch("delta2bbox_clip");
}

namespace {

constexpr int RPN_LOSS_BLOCKS = 1024;

// RPN loss of the whole batch in one pass (anchor_head.py:373-418 loss_single summed over levels and images, with
// the targets of _get_targets_single :172-269 formed on the fly): per anchor row
//   cls:  w * BCEWithLogits(x, t),  t = 1 on sampled positives, w = 1 (pos_weight on positives) on sampled rows
//   box:  SmoothL1_beta(reg - bbox2delta(anchor, gt[assigned-1])) summed over 4, on sampled positives
// and, in the same pass, d(sum)/dx and d(sum)/dreg (the backward only scales them).  Sums are written as per-block
// partials in a fixed grid and reduced in a fixed order afterwards (bitwise reproducible).
__global__ __launch_bounds__(256) void rpn_loss_kernel(const float *__restrict__ cls, const float *__restrict__ reg,
                                                       const float *__restrict__ anchors,
                                                       const float *__restrict__ gts,
                                                       const int64_t *__restrict__ assigned,
                                                       const uint8_t *__restrict__ pos, const uint8_t *__restrict__ neg,
                                                       int64_t rows, int A, int K, Vec4 means, Vec4 stds, float beta,
                                                       float pos_weight, float *__restrict__ partial,
                                                       float *__restrict__ gcls, float *__restrict__ greg)
{
    __shared__ float red[2][4];
    float s_cls = 0.f, s_box = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
        const bool p = pos[i] != 0, n = neg[i] != 0;
        float gx = 0.f;
        float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p || n) {
            const float x = cls[i], t = p ? 1.f : 0.f;
            const float w = (p && pos_weight > 0.f) ? pos_weight : 1.f;
            s_cls += w * (fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
            gx = w * (1.f / (1.f + expf(-x)) - t);
        }
        if (p) {
            const int64_t b = i / A;
            const int a = (int)(i - b * A);
            const float4 an = *reinterpret_cast<const float4 *>(anchors + (int64_t)a * 4);
            const float4 g = *reinterpret_cast<const float4 *>(gts + (b * K + (assigned[i] - 1)) * 4);
            const float px = (an.x + an.z) * 0.5f, py = (an.y + an.w) * 0.5f, pw = an.z - an.x, ph = an.w - an.y;
            const float cx = (g.x + g.z) * 0.5f, cy = (g.y + g.w) * 0.5f, gw = g.z - g.x, gh = g.w - g.y;
            float tgt[4];
            tgt[0] = ((cx - px) / pw - means.v[0]) / stds.v[0];
            tgt[1] = ((cy - py) / ph - means.v[1]) / stds.v[1];
            tgt[2] = (logf(gw / pw) - means.v[2]) / stds.v[2];
            tgt[3] = (logf(gh / ph) - means.v[3]) / stds.v[3];
            const float4 r = *reinterpret_cast<const float4 *>(reg + i * 4);
            const float rr[4] = {r.x, r.y, r.z, r.w};
            float go[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = rr[k] - tgt[k], ad = fabsf(d);
                if (ad < beta) { s_box += 0.5f * ad * ad / beta; go[k] = d / beta; }
                else { s_box += ad - 0.5f * beta; go[k] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
            }
            gr = make_float4(go[0], go[1], go[2], go[3]);
        }
        gcls[i] = gx;
        *reinterpret_cast<float4 *>(greg + i * 4) = gr;
    }
    s_cls = htd::wave_sum(s_cls);
    s_box = htd::wave_sum(s_box);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s_cls; red[1][wave] = s_box; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

}  // namespace

namespace {

constexpr int ROI_LOSS_BLOCKS = 64;

// The two losses of a RoI head stage over all its sample rows in one pass (bbox_heads/bbox_head.py:148-186 with
// losses/cross_entropy_loss.py:9-39, smooth_l1_loss.py:8-26, the class-agnostic regression of the HTD heads):
//   cls:  w_i * (logsumexp(x_i) - x_i[label_i])                                  -> partial[0], and  #(w_i > 0) -> partial[1]
//   box:  sum_k bw_ik * [label_i is a foreground class] * SmoothL1_beta(pred_ik - target_ik)        -> partial[2]
//   acc:  #(argmax x_i == label_i and w_i > 0) (first maximum, like torch.argmax)                  -> partial[3]
// and d(sum)/dx, d(sum)/dpred in the same pass (the backward only scales them by loss weight / avg_factor).  One wavefront
// per row, rows strided over a fixed grid; per-block partials, added in block order by the caller (bitwise reproducible).
__global__ __launch_bounds__(256) void roi_head_loss_kernel(const float *__restrict__ cls, const int64_t *__restrict__ labels,
                                                            const float *__restrict__ lw, const float *__restrict__ pred,
                                                            const float *__restrict__ tgt, const float *__restrict__ bw,
                                                            int64_t n, int NC, int num_fg, float beta,
                                                            float *__restrict__ partial, float *__restrict__ gcls,
                                                            float *__restrict__ gbox)
{
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s_ce = 0.f, s_w = 0.f, s_box = 0.f, s_hit = 0.f;               // lane 0 of the wave keeps the running sums
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t lab = labels[i];
        const float w = lw[i];
        const float *x = cls + i * NC;
        // NC <= 128: two logits per lane
        const int c0 = lane, c1 = lane + 64;
        const float x0 = c0 < NC ? x[c0] : -INFINITY, x1 = c1 < NC ? x[c1] : -INFINITY;
        const float m = htd::wave_max(fmaxf(x0, x1));
        const float e0 = c0 < NC ? expf(x0 - m) : 0.f, e1 = c1 < NC ? expf(x1 - m) : 0.f;
        const float lse = m + logf(htd::wave_sum(e0 + e1));
        // first index of the maximum
        int am = x0 == m ? c0 : (x1 == m ? c1 : 1 << 30);
        for (int o = 32; o > 0; o >>= 1) am = min(am, __shfl_xor(am, o, 64));
        const float xl = (lab >= 0 && lab < NC) ? x[lab] : 0.f;
        if (lane == 0) {
            s_ce += w * (lse - xl);
            s_w += w > 0.f ? 1.f : 0.f;
            s_hit += (w > 0.f && (int64_t)am == lab) ? 1.f : 0.f;
        }
        if (c0 < NC) gcls[i * NC + c0] = w * (expf(x0 - lse) - (c0 == lab ? 1.f : 0.f));
        if (c1 < NC) gcls[i * NC + c1] = w * (expf(x1 - lse) - (c1 == lab ? 1.f : 0.f));
        if (pred) {
            const bool fg = lab >= 0 && lab < num_fg;
            float l = 0.f, g = 0.f;
            if (lane < 4) {
                const float wk = fg ? bw[i * 4 + lane] : 0.f;
                const float d = pred[i * 4 + lane] - tgt[i * 4 + lane], ad = fabsf(d);
                if (ad < beta) { l = wk * (0.5f * ad * ad / beta); g = wk * (d / beta); }
                else { l = wk * (ad - 0.5f * beta); g = wk * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }
                gbox[i * 4 + lane] = g;
            }
            l += __shfl_xor(l, 1, 64);
            l += __shfl_xor(l, 2, 64);
            if (lane == 0) s_box += l;
        }
    }
    if (lane == 0) { red[0][wave] = s_ce; red[1][wave] = s_w; red[2][wave] = s_box; red[3][wave] = s_hit; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float *r = red[threadIdx.x];
        partial[4 * blockIdx.x + threadIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
    }
}

}  // namespace

extern "C" int htd_roi_head_loss_partial_rows(void) { return ROI_LOSS_BLOCKS; }

// cls_score [n][NC] (NC <= 128, last class = background), labels [n] int64, label_weights [n]; bbox_pred / bbox_targets /
// bbox_weights [n][4] (class-agnostic regression; bbox_pred NULL: classification only), num_fg foreground classes.
// -> partial [htd_roi_head_loss_partial_rows()][4] = per-block {sum w*CE, #(w > 0), sum bw*SmoothL1, #correct},
//    grad_cls [n][NC], grad_box [n][4] = derivatives of the two sums.
extern "C" int htd_roi_head_loss(const float *cls_score, const int64_t *labels, const float *label_weights,
                                 const float *bbox_pred, const float *bbox_targets, const float *bbox_weights, int64_t n,
                                 int NC, int num_fg, float beta, float *partial, float *grad_cls, float *grad_box,
                                 void *stream)
{
    HTD_REQUIRE(n > 0 && NC > 0 && NC <= 128 && num_fg >= 0 && beta > 0.f, "roi_head_loss: bad sizes n=%lld NC=%d", (long long)n, NC);
    HTD_REQUIRE(cls_score && labels && label_weights && partial && grad_cls, "roi_head_loss: null pointer");
    HTD_REQUIRE(!bbox_pred || (bbox_targets && bbox_weights && grad_box), "roi_head_loss: null regression pointer");
    hipLaunchKernelGGL(roi_head_loss_kernel, dim3(ROI_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, cls_score, labels,
                       label_weights, bbox_pred, bbox_targets, bbox_weights, n, NC, num_fg, beta, partial, grad_cls, grad_box);
    return htd::check_launch("roi_head_loss");
}

extern "C" int htd_rpn_loss_partial_rows(void) { return RPN_LOSS_BLOCKS; }

extern "C" int htd_rpn_loss(const float *cls, const float *reg, const float *anchors, const float *gts,
                            const int64_t *assigned, const uint8_t *pos, const uint8_t *neg, int B, int A, int K,
                            const float *means4, const float *stds4, float beta, float pos_weight, float *partial,
                            float *grad_cls, float *grad_reg, void *stream)
{
    HTD_REQUIRE(B > 0 && A > 0 && K > 0 && beta > 0.f, "rpn_loss: bad sizes");
    HTD_REQUIRE(cls && reg && anchors && gts && assigned && pos && neg && means4 && stds4 && partial && grad_cls &&
                    grad_reg, "rpn_loss: null pointer");
    Vec4 m, sd;
    for (int k = 0; k < 4; ++k) { m.v[k] = means4[k]; sd.v[k] = stds4[k]; }
    hipLaunchKernelGGL(rpn_loss_kernel, dim3(RPN_LOSS_BLOCKS), dim3(256), 0, (hipStream_t)stream, cls, reg, anchors,
                       gts, assigned, pos, neg, (int64_t)B * A, A, K, m, sd, beta, pos_weight, partial, grad_cls,
                       grad_reg);
    return htd::check_launch("rpn_loss");
}


// ---- map_roi_levels (single_level_roi_extractor.py:32-51, htd_bbox_head.py:129-135) -------------------------------------
// lvl = clamp(floor(log2(sqrt(w * h) / finest_scale + 1e-6)), 0, num_levels - 1) per RoI (idx, x1, y1, x2, y2): the reference's
// five element-wise tensor operations (nine launches here through ATen, three call sites per train step) as one pass, in
// the same fp32 operations in the same order (this file is compiled with -ffp-contract=off).
namespace {
__global__ __launch_bounds__(256) void map_roi_levels_kernel(const float *__restrict__ rois, int64_t *__restrict__ lvls, int64_t n,
                                                             int num_levels, float finest_scale)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *r = rois + 5 * i;
    const float scale = sqrtf((r[3] - r[1]) * (r[4] - r[2]));
    float l = floorf(log2f(scale / finest_scale + 1e-6f));
    l = fminf(fmaxf(l, 0.f), (float)(num_levels - 1));         // (fmaxf / fminf drop a NaN -- a box of negative area -- to level 0:
    lvls[i] = (int64_t)l;                                      //  the result is used as an index, it must stay inside [0, L))
}
}  // namespace

extern "C" int htd_map_roi_levels(const float *rois, int64_t *lvls, int64_t n, int num_levels, float finest_scale, void *stream)
{
    HTD_REQUIRE(n >= 0 && num_levels > 0 && finest_scale > 0.f, "map_roi_levels: bad arguments");
    if (n == 0) return HTD_OK;
    HTD_REQUIRE(rois && lvls, "map_roi_levels: null pointer");
    hipLaunchKernelGGL(map_roi_levels_kernel, dim3((unsigned)htd::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, rois, lvls, n,
                       num_levels, finest_scale);
    return htd::check_launch("map_roi_levels");
}


// ---- RPN head outputs of all pyramid levels <-> the flat per-anchor form (anchor_head.py:172-269 targets / loss, rpn_head.py:
// 78-168 proposals: both flatten every level's (B, A*C, h, w) map with permute(0, 2, 3, 1).reshape and concatenate) -----------
// y[l]: the merged head's output of level l, NHWC [B][P_l][C] with channel c < na the objectness of anchor c and channels
// na + 4 a + j the deltas of anchor a (RPNHead.forward: rpn_cls and rpn_reg as one convolution; C - 5 na padding channels).
//   gather:  cls[b][off_l + p na + a] = y_l[b][p][a],   reg[b][off_l + p na + a][j] = y_l[b][p][na + 4 a + j]
//   scatter: the transpose, padding channels zero -- the gradient of y_l written in full.
// One launch each instead of ~45 slice / clone / concatenate launches per train step.
namespace {
struct HeadLevels {
    const float *y[8];
    float *gy[8];
    int64_t pix[8];        // P_l = h_l * w_l
    int64_t off[8];        // first anchor of level l in an image's flat list
    int64_t t0[9];         // first thread of level l (prefix of B * P_l)
    int L;
};

template <bool SCATTER>
__global__ __launch_bounds__(256) void rpn_heads_kernel(HeadLevels t, float *__restrict__ cls, float *__restrict__ reg, int64_t atot,
                                                        int C, int na)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= t.t0[t.L]) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < t.L && i >= t.t0[k]) l = k;
    const int64_t r = i - t.t0[l], P = t.pix[l];
    const int64_t b = r / P, p = r - b * P;
    const int64_t a0 = b * atot + t.off[l] + p * na;
    if (!SCATTER) {
        const float *src = t.y[l] + r * C;
        for (int a = 0; a < na; ++a) cls[a0 + a] = src[a];
        for (int k = 0; k < 4 * na; ++k) reg[a0 * 4 + k] = src[na + k];
    } else {
        float *dst = t.gy[l] + r * C;
        for (int a = 0; a < na; ++a) dst[a] = cls[a0 + a];
        for (int k = 0; k < 4 * na; ++k) dst[na + k] = reg[a0 * 4 + k];
        for (int c = 5 * na; c < C; ++c) dst[c] = 0.f;
    }
}

int rpn_heads_launch(bool scatter, const float *const *y, float *const *gy, const int64_t *pix, int L, int B, int C, int na,
                     float *cls, float *reg, void *stream, const char *what)
{
    HTD_REQUIRE(L > 0 && L <= 8 && B > 0 && na > 0 && C >= 5 * na && pix && cls && reg, "%s: bad arguments", what);
    HeadLevels t{};
    t.L = L;
    int64_t atot = 0, threads = 0;
    for (int l = 0; l < L; ++l) {
        HTD_REQUIRE(pix[l] > 0 && (scatter ? gy && gy[l] : y && y[l]), "%s: bad level %d", what, l);
        t.y[l] = scatter ? nullptr : y[l];
        t.gy[l] = scatter ? gy[l] : nullptr;
        t.pix[l] = pix[l];
        t.off[l] = atot;
        t.t0[l] = threads;
        atot += pix[l] * na;
        threads += (int64_t)B * pix[l];
    }
    t.t0[L] = threads;
    const unsigned blocks = (unsigned)htd::ceil_div(threads, 256);
    if (scatter)
        hipLaunchKernelGGL(rpn_heads_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, cls, reg, atot, C, na);
    else
        hipLaunchKernelGGL(rpn_heads_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, cls, reg, atot, C, na);
    return htd::check_launch(what);
}
}  // namespace

extern "C" int htd_rpn_heads_gather(const float *const *y, const int64_t *pix, int L, int B, int C, int na, float *cls, float *reg,
                                    void *stream)
{
    return rpn_heads_launch(false, y, nullptr, pix, L, B, C, na, cls, reg, stream, "rpn_heads_gather");
}

extern "C" int htd_rpn_heads_scatter(const float *gcls, const float *greg, float *const *gy, const int64_t *pix, int L, int B,
                                     int C, int na, void *stream)
{
    return rpn_heads_launch(true, nullptr, gy, pix, L, B, C, na, const_cast<float *>(gcls), const_cast<float *>(greg), stream,
                            "rpn_heads_scatter");
}


// ---- StaticSamples from the sampler's slot order (htd_roi_head.py:254-264,292-310; sampling_result.py:40-60) -------------
// order[b][s]: candidate index of slot s (drawn positives first, then drawn negatives: htd_random_sample), counts[b] = (drawn
// positives, drawn negatives).  Candidates are [gts (K, when add_gt) | proposals (P) | never-drawn padding]; assigned[b][c] is
// the 1-based gt index of candidate c (<= 0: none).  One pass writes what ~14 gather / compare / concatenate launches made:
//   boxes = candidate box * (slot used), valid, is_pos, the assigned gt's box and label per slot, pos_is_gt.
namespace {
__global__ __launch_bounds__(256) void static_samples_finish_kernel(
    const float *__restrict__ gts, const uint8_t *__restrict__ gvalid, const int64_t *__restrict__ glabels,
    const float *__restrict__ props, const int64_t *__restrict__ assigned, const int64_t *__restrict__ order,
    const int64_t *__restrict__ counts, int B, int K, int P, int A, int S, int add_gt, float *__restrict__ boxes,
    uint8_t *__restrict__ valid, uint8_t *__restrict__ is_pos, float *__restrict__ pos_gt_boxes, int64_t *__restrict__ pos_gt_labels,
    uint8_t *__restrict__ pos_is_gt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * S) return;
    const int b = i / S, s = i - b * S;
    const int64_t npos = counts[2 * b], n = npos + counts[2 * b + 1];
    const bool v = s < n, ip = s < npos;
    const int64_t c = order[i];
    const int koff = add_gt ? K : 0;
    float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
    bool from_gt = false;
    if (add_gt && c < K) {
        box = *reinterpret_cast<const float4 *>(gts + ((int64_t)b * K + c) * 4);
        from_gt = gvalid[(int64_t)b * K + c] != 0;
    } else if (c - koff < P) {
        box = *reinterpret_cast<const float4 *>(props + ((int64_t)b * P + (c - koff)) * 4);
    }
    const float m = v ? 1.f : 0.f;          // (the tensor form multiplies by the mask: a negative coordinate of an unused slot is -0)
    *reinterpret_cast<float4 *>(boxes + (int64_t)i * 4) = make_float4(box.x * m, box.y * m, box.z * m, box.w * m);
    valid[i] = v;
    is_pos[i] = ip;
    int64_t g = assigned[(int64_t)b * A + c] - 1;
    g = g < 0 ? 0 : g;
    *reinterpret_cast<float4 *>(pos_gt_boxes + (int64_t)i * 4) = *reinterpret_cast<const float4 *>(gts + ((int64_t)b * K + g) * 4);
    pos_gt_labels[i] = glabels[(int64_t)b * K + g];
    pos_is_gt[i] = from_gt && ip;
}
}  // namespace

extern "C" int htd_static_samples_finish(const float *gts, const uint8_t *gvalid, const int64_t *glabels, const float *props,
                                         const int64_t *assigned, const int64_t *order, const int64_t *counts, int B, int K, int P,
                                         int A, int S, int add_gt, float *boxes, uint8_t *valid, uint8_t *is_pos,
                                         float *pos_gt_boxes, int64_t *pos_gt_labels, uint8_t *pos_is_gt, void *stream)
{
    HTD_REQUIRE(B > 0 && K > 0 && P >= 0 && S > 0 && A >= (add_gt ? K : 0) + P, "static_samples_finish: bad sizes");
    HTD_REQUIRE(gts && gvalid && glabels && (props || P == 0) && assigned && order && counts && boxes && valid && is_pos &&
                    pos_gt_boxes && pos_gt_labels && pos_is_gt, "static_samples_finish: null pointer");
    hipLaunchKernelGGL(static_samples_finish_kernel, dim3((unsigned)htd::ceil_div((int64_t)B * S, 256)), dim3(256), 0,
                       (hipStream_t)stream, gts, gvalid, glabels, props, assigned, order, counts, B, K, P, A, S, add_gt, boxes, valid,
                       is_pos, pos_gt_boxes, pos_gt_labels, pos_is_gt);
    return htd::check_launch("static_samples_finish");
}
