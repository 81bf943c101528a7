/*
 * oracle/c/htd_oracle_ops.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the three native operators the HTD hot path
 * calls through mmcv.ops.  Their kernels live in the un-vendored third-party
 * dependency mmcv-full==1.2.1 (pinned by /root/reference/README.md:11), which
 * is absent from /root/reference, so this file restates the *published*
 * algorithm of that release and is anchored on the reference's call sites:
 *
 *   RoIAlign      roi_extractors/base_roi_extractor.py:49-56,
 *                 single_level_roi_extractor.py:93, adaptative_roi_extractor.py:72,87
 *                 native signature: build/lib/mmdet/ops/roi_align/roi_align.py:28-30,67-71
 *   nms/soft_nms  dense_heads/rpn_head.py:166-167, core/post_processing/bbox_nms.py:65
 *                 native signature: build/lib/mmdet/ops/nms/nms_wrapper.py:52-55,62-116
 *   DeformConv    backbones/resnet.py:186-194
 *                 native signature: build/lib/mmdet/ops/dcn/deform_conv.py:51-56,75-91
 *
 * PARITY UNPINNED for these three operators: the reference holds no numeric
 * golden vector for them.  They are cross-checked
 * against closed-form cases in tests/test_oracle_ops.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Everything is single-threaded scalar C on purpose.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* RoIAlign (aligned=True/False, avg pooling, adaptive sampling grid)  */
/* features NCHW fp32, rois (n,5) = [batch_idx, x1, y1, x2, y2]        */
/* ------------------------------------------------------------------ */

static float bilinear_at(const float *plane, int H, int W, float y, float x)
{
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    float ly = y - (float)y_low, lx = x - (float)x_low;
    float hy = 1.f - ly, hx = 1.f - lx;
    float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * plane[y_low * W + x_low] + w2 * plane[y_low * W + x_high] +
           w3 * plane[y_high * W + x_low] + w4 * plane[y_high * W + x_high];
}

static void bilinear_scatter(float *plane, int H, int W, float y, float x, float g)
{
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    float ly = y - (float)y_low, lx = x - (float)x_low;
    float hy = 1.f - ly, hx = 1.f - lx;
    plane[y_low * W + x_low] += g * (hy * hx);
    plane[y_low * W + x_high] += g * (hy * lx);
    plane[y_high * W + x_low] += g * (ly * hx);
    plane[y_high * W + x_high] += g * (ly * lx);
}

typedef struct {
    float start_h, start_w, bin_h, bin_w;
    int grid_h, grid_w, batch;
} roi_geom;

static roi_geom roi_geometry(const float *roi, float scale, int ph, int pw, int sampling_ratio,
                             int aligned)
{
    roi_geom g;
    float off = aligned ? 0.5f : 0.f;
    g.batch = (int)roi[0];
    g.start_w = roi[1] * scale - off;
    g.start_h = roi[2] * scale - off;
    float end_w = roi[3] * scale - off, end_h = roi[4] * scale - off;
    float rw = end_w - g.start_w, rh = end_h - g.start_h;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    g.bin_h = rh / (float)ph;
    g.bin_w = rw / (float)pw;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
    return g;
}

int oracle_roi_align_fwd(const float *feat, const float *rois, float *out, int n_rois, int C, int H,
                         int W, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned)
{
    for (int n = 0; n < n_rois; ++n) {
        roi_geom g = roi_geometry(rois + 5 * n, spatial_scale, ph, pw, sampling_ratio, aligned);
        int cnt_i = g.grid_h * g.grid_w;
        float count = (float)(cnt_i > 1 ? cnt_i : 1);
        for (int c = 0; c < C; ++c) {
            const float *plane = feat + ((size_t)g.batch * C + c) * H * W;
            for (int i = 0; i < ph; ++i)
                for (int j = 0; j < pw; ++j) {
                    float acc = 0.f;
                    for (int iy = 0; iy < g.grid_h; ++iy) {
                        float y = g.start_h + i * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
                        for (int ix = 0; ix < g.grid_w; ++ix) {
                            float x = g.start_w + j * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
                            acc += bilinear_at(plane, H, W, y, x);
                        }
                    }
                    out[(((size_t)n * C + c) * ph + i) * pw + j] = acc / count;
                }
        }
    }
    return 0;
}

/* grad_in must be zero-initialised by the caller (B,C,H,W). */
int oracle_roi_align_bwd(const float *grad_out, const float *rois, float *grad_in, int n_rois, int C,
                         int H, int W, int ph, int pw, float spatial_scale, int sampling_ratio,
                         int aligned)
{
    for (int n = 0; n < n_rois; ++n) {
        roi_geom g = roi_geometry(rois + 5 * n, spatial_scale, ph, pw, sampling_ratio, aligned);
        int cnt_i = g.grid_h * g.grid_w;
        float count = (float)(cnt_i > 1 ? cnt_i : 1);
        for (int c = 0; c < C; ++c) {
            float *plane = grad_in + ((size_t)g.batch * C + c) * H * W;
            for (int i = 0; i < ph; ++i)
                for (int j = 0; j < pw; ++j) {
                    float go = grad_out[(((size_t)n * C + c) * ph + i) * pw + j] / count;
                    for (int iy = 0; iy < g.grid_h; ++iy) {
                        float y = g.start_h + i * g.bin_h + (iy + .5f) * g.bin_h / (float)g.grid_h;
                        for (int ix = 0; ix < g.grid_w; ++ix) {
                            float x = g.start_w + j * g.bin_w + (ix + .5f) * g.bin_w / (float)g.grid_w;
                            bilinear_scatter(plane, H, W, y, x, go);
                        }
                    }
                }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Hard NMS: sort by score (descending, stable => lower index first on  */
/* ties), greedy suppression of IoU > thr.  keep_out receives indices   */
/* into the ORIGINAL box array in descending-score order.  Returns k.   */
/* ------------------------------------------------------------------ */

typedef struct { float s; int64_t i; } sc_idx;
static int cmp_desc(const void *a, const void *b)
{
    const sc_idx *x = (const sc_idx *)a, *y = (const sc_idx *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

int64_t oracle_nms(const float *boxes, const float *scores, int64_t n, float iou_thr, int offset,
                   int64_t *keep_out)
{
    if (n == 0) return 0;
    sc_idx *ord = (sc_idx *)malloc(sizeof(sc_idx) * n);
    float *area = (float *)malloc(sizeof(float) * n);
    uint8_t *dead = (uint8_t *)calloc(n, 1);
    for (int64_t i = 0; i < n; ++i) {
        ord[i].s = scores[i];
        ord[i].i = i;
        area[i] = (boxes[4 * i + 2] - boxes[4 * i] + offset) * (boxes[4 * i + 3] - boxes[4 * i + 1] + offset);
    }
    qsort(ord, n, sizeof(sc_idx), cmp_desc);
    int64_t k = 0;
    for (int64_t a = 0; a < n; ++a) {
        if (dead[a]) continue;
        int64_t i = ord[a].i;
        keep_out[k++] = i;
        const float *bi = boxes + 4 * i;
        for (int64_t b = a + 1; b < n; ++b) {
            if (dead[b]) continue;
            int64_t j = ord[b].i;
            const float *bj = boxes + 4 * j;
            float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
            float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
            float w = fmaxf(0.f, xx2 - xx1 + offset), h = fmaxf(0.f, yy2 - yy1 + offset);
            float inter = w * h;
            float ovr = inter / (area[i] + area[j] - inter);
            if (ovr > iou_thr) dead[b] = 1;
        }
    }
    free(ord); free(area); free(dead);
    return k;
}

/* ------------------------------------------------------------------ */
/* Soft-NMS (method 0 naive, 1 linear, 2 gaussian), sequential.         */
/* dets_out (n,5) receives [x1,y1,x2,y2,new_score]; inds_out the        */
/* original indices; returns number kept.                               */
/* ------------------------------------------------------------------ */

int64_t oracle_soft_nms(const float *boxes, const float *scores, int64_t n, float iou_thr, float sigma,
                        float min_score, int method, int offset, float *dets_out, int64_t *inds_out)
{
    float *x1 = (float *)malloc(sizeof(float) * n * 6);
    float *y1 = x1 + n, *x2 = y1 + n, *y2 = x2 + n, *sc = y2 + n, *ar = sc + n;
    for (int64_t i = 0; i < n; ++i) {
        x1[i] = boxes[4 * i]; y1[i] = boxes[4 * i + 1]; x2[i] = boxes[4 * i + 2]; y2[i] = boxes[4 * i + 3];
        sc[i] = scores[i];
        ar[i] = (x2[i] - x1[i] + offset) * (y2[i] - y1[i] + offset);
        inds_out[i] = i;
    }
#define SWAPF(a, p, q) do { float t_ = a[p]; a[p] = a[q]; a[q] = t_; } while (0)
#define SWAPALL(p, q) do { SWAPF(x1, p, q); SWAPF(y1, p, q); SWAPF(x2, p, q); SWAPF(y2, p, q); \
        SWAPF(sc, p, q); SWAPF(ar, p, q); int64_t t2_ = inds_out[p]; inds_out[p] = inds_out[q]; inds_out[q] = t2_; } while (0)
    int64_t nb = n;
    for (int64_t i = 0; i < nb; ++i) {
        float max_score = sc[i];
        int64_t max_pos = i;
        for (int64_t pos = i + 1; pos < nb; ++pos)
            if (max_score < sc[pos]) { max_score = sc[pos]; max_pos = pos; }
        SWAPALL(i, max_pos);
        float ix1 = x1[i], iy1 = y1[i], ix2 = x2[i], iy2 = y2[i], iarea = ar[i];
        int64_t pos = i + 1;
        while (pos < nb) {
            float xx1 = fmaxf(ix1, x1[pos]), yy1 = fmaxf(iy1, y1[pos]);
            float xx2 = fminf(ix2, x2[pos]), yy2 = fminf(iy2, y2[pos]);
            float w = fmaxf(0.f, xx2 - xx1 + offset), h = fmaxf(0.f, yy2 - yy1 + offset);
            float inter = w * h;
            float ovr = inter / (iarea + ar[pos] - inter);
            float weight = 1.f;
            if (method == 0) { if (ovr >= iou_thr) weight = 0.f; }
            else if (method == 1) { if (ovr >= iou_thr) weight = 1.f - ovr; }
            else if (method == 2) { weight = expf(-(ovr * ovr) / sigma); }
            sc[pos] *= weight;
            if (sc[pos] < min_score) {
                SWAPALL(pos, nb - 1);
                nb -= 1;
                pos -= 1;
            }
            pos += 1;
        }
    }
    for (int64_t i = 0; i < nb; ++i) {
        dets_out[5 * i] = x1[i]; dets_out[5 * i + 1] = y1[i]; dets_out[5 * i + 2] = x2[i];
        dets_out[5 * i + 3] = y2[i]; dets_out[5 * i + 4] = sc[i];
    }
    free(x1);
    return nb;
#undef SWAPF
#undef SWAPALL
}

/* ------------------------------------------------------------------ */
/* Deformable convolution v1/v2 forward (mask may be NULL => v1).       */
/* input (B,C,H,W), offset (B, 2*dg*kh*kw, Ho, Wo) laid out             */
/* [dy0,dx0,dy1,dx1,...], mask (B, dg*kh*kw, Ho, Wo), weight            */
/* (Co, C/groups, kh, kw), out (B,Co,Ho,Wo).  Zero-padding bilinear.    */
/* ------------------------------------------------------------------ */

static float dcn_bilinear(const float *plane, int H, int W, float h, float w)
{
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = plane[h_low * W + w_low];
    if (h_low >= 0 && w_high <= W - 1) v2 = plane[h_low * W + w_high];
    if (h_high <= H - 1 && w_low >= 0) v3 = plane[h_high * W + w_low];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = plane[h_high * W + w_high];
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

int oracle_deform_conv_fwd(const float *in, const float *offset, const float *mask, const float *weight,
                           float *out, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw,
                           int padh, int padw, int dilh, int dilw, int groups, int dgroups)
{
    int Ho = (H + 2 * padh - (dilh * (kh - 1) + 1)) / sh + 1;
    int Wo = (W + 2 * padw - (dilw * (kw - 1) + 1)) / sw + 1;
    int Cg = C / groups, Cog = Co / groups, cpd = C / dgroups;
    float *col = (float *)malloc(sizeof(float) * (size_t)C * kh * kw);
    for (int b = 0; b < B; ++b)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
                for (int c = 0; c < C; ++c) {
                    int dg = c / cpd;
                    const float *plane = in + ((size_t)b * C + c) * H * W;
                    for (int i = 0; i < kh; ++i)
                        for (int j = 0; j < kw; ++j) {
                            int k = i * kw + j;
                            size_t obase = (((size_t)b * dgroups + dg) * 2 * kh * kw);
                            float oh = offset[((obase + 2 * k) * Ho + ho) * Wo + wo];
                            float ow = offset[((obase + 2 * k + 1) * Ho + ho) * Wo + wo];
                            float hi = (float)(ho * sh - padh + i * dilh) + oh;
                            float wi = (float)(wo * sw - padw + j * dilw) + ow;
                            float v = 0.f;
                            if (hi > -1 && wi > -1 && hi < H && wi < W) v = dcn_bilinear(plane, H, W, hi, wi);
                            if (mask)
                                v *= mask[((((size_t)b * dgroups + dg) * kh * kw + k) * Ho + ho) * Wo + wo];
                            col[(c * kh + i) * kw + j] = v;
                        }
                }
                for (int co = 0; co < Co; ++co) {
                    int g = co / Cog;
                    const float *wrow = weight + (size_t)co * Cg * kh * kw;
                    const float *crow = col + (size_t)g * Cg * kh * kw;
                    float acc = 0.f;
                    for (int t = 0; t < Cg * kh * kw; ++t) acc += wrow[t] * crow[t];
                    out[(((size_t)b * Co + co) * Ho + ho) * Wo + wo] = acc;
                }
            }
    free(col);
    return 0;
}
