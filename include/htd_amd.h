/*
 * htd_amd.h -- C ABI of libhtd_amd.so: the MI355X (gfx950) operator layer under the
 * HTD detection hot path.  This is the drop-in boundary: every entry point replaces one
 * native operator that the reference reaches through mmcv.ops / ATen.
 *
 * Conventions
 *   - plain pointers + explicit sizes, no torch types; all pointers are DEVICE pointers
 *     unless a parameter is documented as host;
 *   - activations are NHWC fp32: feat[b][y][x][c]; RoI features are [n][ph][pw][c];
 *     conv weights are KRSC: w[co][kh][kw][ci] (= torch channels_last of (co,ci,kh,kw));
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued there and the call
 *     returns without synchronising; the library never allocates or frees caller-visible
 *     memory (outputs and workspaces are caller-owned, as in the reference wrappers
 *     build/lib/mmdet/ops/dcn/deform_conv.py:37-41, roi_align/roi_align.py:36);
 *   - return value 0 = ok, non-zero = error; htd_last_error() gives a thread-local
 *     message.  The Python side raises ValueError / RuntimeError from it, like the
 *     reference wrappers (deform_conv.py:27-29, roi_align.py:39-40).  Never aborts.
 */
#ifndef HTD_AMD_H
#define HTD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HTD_OK 0
#define HTD_ERR_ARG 1
#define HTD_ERR_LAUNCH 2

const char *htd_last_error(void);
/* ABI version, bumped on any signature change (added entry points do not bump it).  A binding compares
 * htd_abi_version() of the loaded library with the HTD_ABI_VERSION of the header it was written against. */
#define HTD_ABI_VERSION 3
int htd_abi_version(void);

/* ------------------------------------------------------------------------------------
 * RoIAlign (avg pooling, adaptive sampling grid when sampling_ratio == 0).
 * Replaces mmcv.ops.RoIAlign as built at roi_extractors/base_roi_extractor.py:49-56 and
 * called at single_level_roi_extractor.py:93, adaptative_roi_extractor.py:72,87; native
 * signature of the reference era: roi_align_ext.forward_v2(features, rois, spatial_scale,
 * out_h, out_w, sample_num, aligned) build/lib/mmdet/ops/roi_align/roi_align.py:28-30,
 * backward_v2(...) :67-71.
 *   feat  [B][H][W][C]      rois [n][5] = (batch_idx, x1, y1, x2, y2)
 *   out   [n][ph][pw][C]    written in full
 *   roi_level (may be NULL): int64 [n]; when given, only RoIs with roi_level[i] == level
 *     are processed and the other rows of `out` are left untouched -- one (N,ph,pw,C)
 *     tensor is filled level by level with no index lists and no host synchronisation
 *     (SingleRoIExtractor.forward :81-99 does nonzero() + scatter per level).
 * bwd: grad_feat must be zero-initialised (or hold a running sum); accumulates atomically.
 * ---------------------------------------------------------------------------------- */
int htd_roi_align_fwd(const float *feat, const float *rois, const int64_t *roi_level, int level,
                      float *out, int64_t n, int B, int C, int H, int W, int ph, int pw, float spatial_scale,
                      int sampling_ratio, int aligned, void *stream);
int htd_roi_align_bwd(const float *grad_out, const float *rois, const int64_t *roi_level, int level,
                      float *grad_feat, int64_t n, int B, int C, int H, int W, int ph, int pw,
                      float spatial_scale, int sampling_ratio, int aligned, void *stream);
/* Gather form of the same backward: a wavefront owns a strip of 8 pixels of one feature-map row and sums the RoIs that
 * cover it in ascending RoI order -- no float atomics (the scatter form is bound by their 1.3 TB/s and is not bit-stable
 * from run to run), every pixel written once.  accumulate = 0: grad_feat is overwritten everywhere (no memset needed);
 * 1: added to, strips no RoI covers are left alone.  workspace: htd_roi_align_bwd_gather_workspace_bytes(n). */
/* All pyramid levels of SingleRoIExtractor.forward (single_level_roi_extractor.py:81-99) in one launch: RoI i is pooled
 * from feats[roi_level[i]] ([B][H[l]][W[l]][C], spatial_scale scales[l], l < L <= 8; other levels: zeros).  feats, H, W,
 * scales are HOST arrays of L entries. */
int htd_roi_align_levels_fwd(const float *const *feats, const int *H, const int *W, const float *scales, int L,
                             const float *rois, const int64_t *roi_level, float *out, int64_t n, int B, int C, int ph, int pw,
                             int sampling_ratio, int aligned, void *stream);
/* The same, and max |out| is left in *amax_out (device scalar, zero or an earlier maximum on entry) for the H2 launches of the
 * FC layer the tiles go into (htd_conv2d_fwd_x3h / htd_conv2d_bwd_weight_h2). */
int htd_roi_align_levels_fwd_amax(const float *const *feats, const int *H, const int *W, const float *scales, int L,
                                  const float *rois, const int64_t *roi_level, float *out, int64_t n, int B, int C, int ph,
                                  int pw, int sampling_ratio, int aligned, float *amax_out, void *stream);
/* Gather-form RoIAlign backward of all pyramid levels of a SingleRoIExtractor in one launch (coarsest level first, so the
 * long per-strip RoI lists of the small maps overlap the many short strips of the large ones).  grad_feats[l] == NULL
 * skips level l; accumulate[l] != 0 adds into a map another consumer has already written.
 * Every map equals the per-level entry point's bit for bit.
 * workspace: L * htd_roi_align_bwd_gather_workspace_bytes(n) bytes.  Bit-stable (no atomics). */
int htd_roi_align_levels_bwd_gather(const float *grad_out, const float *rois, const int64_t *roi_level,
                                    float *const *grad_feats, const int *H, const int *W, const float *scales,
                                    const int *accumulate, int L, int64_t n, int B, int C, int ph, int pw,
                                    int sampling_ratio, int aligned, void *workspace, void *stream);
/* AdptRoIExtractor (BA, roi_extractors/adaptative_roi_extractor.py:66-76) pools EVERY RoI from EVERY level: the four RoIAlign
 * calls of the reference as one launch each way.  outs[l] / grad_outs[l]: level l's (n, ph, pw, C) tensor.  The backward is the
 * gather form above with roi_level == NULL (every RoI on every level); workspace: L * htd_roi_align_bwd_gather_workspace_bytes(n). */
int htd_roi_align_all_levels_fwd(const float *const *feats, const int *H, const int *W, const float *scales, int L,
                                 const float *rois, float *const *outs, int64_t n, int B, int C, int ph, int pw,
                                 int sampling_ratio, int aligned, void *stream);
int htd_roi_align_all_levels_bwd_gather(const float *const *grad_outs, const float *rois, float *const *grad_feats,
                                        const int *H, const int *W, const float *scales, const int *accumulate, int L,
                                        int64_t n, int B, int C, int ph, int pw, int sampling_ratio, int aligned,
                                        void *workspace, void *stream);
/* Both level-fused gather backwards with the bins of a RoI folded along y ONCE per (RoI, map row) by a pass of its own
 * (the strips of a row used to fetch and fold the same 7-14 bin vectors each): bit-identical gradient maps.
 * fold_ws: htd_roi_align_fold_workspace_bytes(n, H, L, pw, C) bytes = n * sum(H) * pw * C floats. */
int64_t htd_roi_align_fold_workspace_bytes(int64_t n, const int *H, int L, int pw, int C);
int htd_roi_align_levels_bwd_gather_folded(const float *grad_out, const float *rois, const int64_t *roi_level,
                                           float *const *grad_feats, const int *H, const int *W, const float *scales,
                                           const int *accumulate, int L, int64_t n, int B, int C, int ph, int pw,
                                           int sampling_ratio, int aligned, void *workspace, void *fold_ws, void *stream);
int htd_roi_align_all_levels_bwd_gather_folded(const float *const *grad_outs, const float *rois, float *const *grad_feats,
                                               const int *H, const int *W, const float *scales, const int *accumulate,
                                               int L, int64_t n, int B, int C, int ph, int pw, int sampling_ratio,
                                               int aligned, void *workspace, void *fold_ws, void *stream);
int64_t htd_roi_align_bwd_gather_workspace_bytes(int64_t n);
int htd_roi_align_bwd_gather(const float *grad_out, const float *rois, const int64_t *roi_level, int level,
                             float *grad_feat, int64_t n, int B, int C, int H, int W, int ph, int pw,
                             float spatial_scale, int sampling_ratio, int aligned, int accumulate, void *workspace,
                             void *stream);

/* ------------------------------------------------------------------------------------
 * Hard NMS on boxes ALREADY SORTED by descending score (ties: lower original index
 * first).  Replaces nms_ext.nms(dets, iou_thr) build/lib/mmdet/ops/nms/nms_wrapper.py:52-55
 * as used by batched_nms at dense_heads/rpn_head.py:166-167 and
 * core/post_processing/bbox_nms.py:65.  Suppression rule: inter / union > iou_thr with an
 * IEEE fp32 division (the CPU path's arithmetic), offset in {0,1}.
 *   boxes [n][4] sorted;  keep_mask [n] uint8 out (1 = survivor);
 *   workspace: htd_nms_workspace_bytes(n) bytes.
 * Batched form: `segments` problems stored back to back; seg_offsets[segments+1] (device,
 * int64) gives each problem's [begin,end) row range; one launch handles them all, no host
 * synchronisation (per-image RPN NMS, rpn_head.py:78-168).
 * ---------------------------------------------------------------------------------- */
int64_t htd_nms_workspace_bytes(int64_t n_total);
int htd_nms_sorted(const float *boxes, uint8_t *keep_mask, int64_t n, float iou_thr, int offset,
                   void *workspace, void *stream);
int htd_nms_sorted_batched(const float *boxes, const int64_t *seg_offsets, int segments,
                           int64_t n_total, int64_t max_seg, uint8_t *keep_mask, float iou_thr,
                           int offset, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------
 * Soft-NMS over independent segments ((image, class) groups), on the device.  Replaces
 * nms_ext.soft_nms(dets_cpu, iou_thr, method_code, sigma, min_score)
 * build/lib/mmdet/ops/nms/nms_wrapper.py:62-116 (sequential, CPU-only in mmcv) as selected by
 * configs/htd/htd_resnet101_2x.py:298 through multiclass_nms (core/post_processing/bbox_nms.py:65).
 *   boxes [n][4]; scores [n] in/out (decayed in place); seg_offsets [segments+1] device int64;
 *   rank [n] out: selection round within the segment (>= 0, kept) or < 0 (dropped below min_score);
 *   method 0 naive, 1 linear, 2 gaussian.
 * ---------------------------------------------------------------------------------- */
int htd_soft_nms_segments(const float *boxes, float *scores, const int64_t *seg_offsets, int segments,
                          int64_t n_total, int *rank, float iou_thr, float sigma, float min_score,
                          int method, int offset, void *stream);

/* ------------------------------------------------------------------------------------
 * Segmented exact top-k, sorted: for every segment of `keys` its k largest keys in descending order, equal keys by
 * ascending position -- `scores.sort(descending=True)` + `[:nms_pre]` per level and image of RPNHead._get_bboxes_single
 * (dense_heads/rpn_head.py:122-133) and the "n smallest random keys" of RandomSampler.random_choice
 * (core/bbox/samplers/random_sampler.py:33-56) in six launches for all segments of a call.
 *   segs [S][4] device int64: (first key, length, k, first output slot), 0 <= k <= min(length, 2048);
 *   chunk_tab [nchunks][2] device int32: (segment, chunk inside it), the ceil(length / 4096) chunks of a segment
 *   consecutive and ascending;  out_idx: position inside the segment;  out_val: the key.
 * ---------------------------------------------------------------------------------- */
int64_t htd_segmented_topk_workspace_bytes(int S, int64_t nchunks);
int htd_segmented_topk(const float *keys, const int64_t *segs, const int32_t *chunk_tab, int S, int64_t nchunks,
                       int64_t *out_idx, float *out_val, void *workspace, void *stream);
/* RandomSampler for a batch (core/bbox/samplers/base_sampler.py:34-101, random_sampler.py:33-78) without a host round trip:
 * a uniformly random subset of n candidates is the n candidates with the smallest i.i.d. keys.  assigned [B][A] =
 * AssignResult.gt_inds (> 0 positive, 0 negative, < 0 ignored), keys [B][A] in [0, 1).  Draws min(max_pos, #positives)
 * positives, then min(num - drawn positives, neg_pos_ub bound, #negatives) negatives.  segs / chunk_tab: the device tables of
 * htd_segmented_topk for the 2B segments (b*A, A, kpos, b*kpos) and (B*A + b*A, A, kneg, B*kpos + b*kneg), kpos =
 * min(max_pos, A), kneg = min(num, A), num <= 2048.  pos_mask / neg_mask [B][A] bytes 0/1; counts [B][2] drawn (pos, neg);
 * order [B][slots] (may be NULL): drawn positives by ascending index, then drawn negatives by ascending index (the row order
 * of SamplingResult.bboxes, sampling_result.py:23-38), unused slots 0. */
int64_t htd_random_sample_workspace_bytes(int B, int64_t A, int64_t nchunks);
int htd_random_sample(const int64_t *assigned, const float *keys, int B, int64_t A, int num, int max_pos, float neg_pos_ub,
                      const int64_t *segs, const int32_t *chunk_tab, int64_t nchunks, unsigned char *pos_mask,
                      unsigned char *neg_mask, int64_t *counts, int64_t *order, int slots, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------
 * _fuse_global (htd_roi_head.py:133-141 == htd_bbox_head.py:147-155):
 *   out[i][p][c] = roi_feats[i][p][c] + global_feat[img(i)][c],  img(i) = (int)rois[i][0]
 * optionally + alpha * extra[i][p][c]  (x_reg + g + alpha*enhanced, htd_bbox_head.py:163,184).
 * bwd_global: grad_global[b][c] += sum over RoIs of image b and positions of grad[i][p][c].
 * ---------------------------------------------------------------------------------- */
int htd_fuse_global_fwd(const float *roi_feats, const float *rois, const float *global_feat,
                        const float *extra, float alpha, float *out, int64_t n, int P, int C,
                        int B, void *stream);
/* The two input batches of the HTD classification FCs in one pass over roi_feats (htd_bbox_head.py:198,201):
 * both[0..n) = roi_feats, both[n..2n) = roi_feats + global_feat[image of the RoI].  both: [2n][P][C] floats. */
int htd_plain_and_fused_fwd(const float *roi_feats, const float *rois, const float *global_feat, float *both,
                            int64_t n, int P, int C, int B, void *stream);
/* The two above, also leaving the largest magnitude they store in *amax_out (device scalar, zero or an earlier maximum on
 * entry): the `amax` of the H2 launches of the FC layer that reads the tiles (htd_conv2d_fwd_x3h, htd_conv2d_bwd_weight_h2). */
int htd_fuse_global_fwd_amax(const float *roi_feats, const float *rois, const float *global_feat, const float *extra,
                             float alpha, float *out, int64_t n, int P, int C, int B, float *amax_out, void *stream);
int htd_plain_and_fused_fwd_amax(const float *roi_feats, const float *rois, const float *global_feat, float *both,
                                 int64_t n, int P, int C, int B, float *amax_out, void *stream);
int htd_fuse_global_bwd_global(const float *grad, const float *rois, float *grad_global,
                               int64_t n, int P, int C, int B, void *stream);

/* ------------------------------------------------------------------------------------
 * BA fusion (AdptRoIExtractor.forward adaptative_roi_extractor.py:76-91), given the four
 * per-level RoIAlign outputs lvl[l] [n][P][C], the P2 border RoIAlign `border` (may alias
 * lvl[0]) and the per-level attention logits att [L][n]:
 *   w = softmax_l(att);  out = sum_l w[l][i]*lvl[l][i] + border[i] * ring(p)
 * ring(p) = 1 on the outermost `edge` rows/cols of the ph x pw window, else 0.
 * bwd: grad_lvl[l] = w[l]*g (+ ring*g added into grad_lvl[0] when border aliases lvl[0]),
 *      grad_att[l][i] = w[l][i] * (d[l][i] - sum_m w[m][i] d[m][i]),  d[l][i] = <g[i], lvl[l][i]>.
 * ---------------------------------------------------------------------------------- */
int htd_ba_fuse_fwd(const float *const *lvl, int L, const float *border, const float *att,
                    float *out, int64_t n, int ph, int pw, int C, int edge, void *stream);
int htd_ba_fuse_bwd(const float *const *lvl, int L, const float *att, const float *grad_out,
                    float *const *grad_lvl, float *grad_border, float *grad_att, int64_t n, int ph,
                    int pw, int C, int edge, void *stream);

/* ------------------------------------------------------------------------------------
 * Dense NHWC fp32 convolution on the f32 matrix cores (v_mfma_f32_32x32x2_f32), implicit GEMM with
 * the fused epilogue  y = act(conv(x, w) + bias + residual).  Replaces the ATen/cuDNN convolutions and
 * GEMMs of ResNet/FPN/RPN/SFA/HTD-reg and every nn.Linear of the heads (backbones/resnet.py:260-300,
 * necks/fpn.py:165-216, dense_heads/rpn_head.py:37-43, global_context_head.py:382-392,
 * htd_bbox_head.py:164,186,192,194,216,227-228, convfc_bbox_head.py:147-172).
 *   x [B][H][W][Ci]   w [Co][kh][kw][Ci]   bias [Co] or NULL   residual [B][Ho][Wo][Co] or NULL;
 *   res_h, res_w > 0: residual is a coarser [B][res_h][res_w][Co] map added through nearest-neighbour up-sampling
 *   (the FPN top-down `laterals[i-1] += F.interpolate(laterals[i], mode='nearest')`, necks/fpn.py:176-189)
 *   y [B][Ho][Wo][Co];  relu in {0,1};  Ci % 8 == 0 (the 3-channel stem input is padded to 8).
 *   A Linear layer is the 1x1 case with H = rows, W = 1.
 * workspace (fwd / bwd_data): optional split-K scratch of htd_conv2d_workspace_bytes(M, Co, Ci, kh, kw) bytes
 *            (M = output pixels, Co/Ci = output / reduction channels of that GEMM); problems too small to fill
 *            256 CUs are split along K and summed in a second pass.  NULL disables split-K.
 * bwd_data:  gx = conv_transpose(gy, w), given wT = htd_conv2d_flip_weights(w) ([Ci][kh][kw][Co], taps
 *            reversed).  mask_src (may be NULL, [B][H][W][Ci]): gx is zeroed where mask_src <= 0, i.e. the
 *            backward of the ReLU that produced the conv input is fused into this epilogue.  Co % 8 == 0.
 * bwd_weight: gw[co][kh][kw][ci] = sum_pixels gy * x; deterministic split-K through `workspace`
 *            (htd_conv2d_wgrad_workspace_bytes); Ci % 4 == 0.  gbias (may be NULL, [Co]): the bias gradient
 *            sum_pixels gy is accumulated by the same kernel (the tiles of the first N column see every gy element).
 * bias_grad_relu_mask: gbias[c] = sum_rows gm[r][c] with gm = g * (y > 0) written out when y != NULL
 *            (gm = g, nothing written, when y == NULL); workspace >= 2048*C*4 bytes.  gbias == NULL (y required):
 *            ReLU mask only.
 * ---------------------------------------------------------------------------------- */
int64_t htd_conv2d_workspace_bytes(int64_t M, int Co, int Ci, int kh, int kw);
/* Tile table of conv_igemm_kernel (htd_conv2d_fwd / htd_conv2d_bwd_data): problem (M = B*Ho*Wo, Co, Ci, taps = kh*kw,
 * epi: bit 0 residual / accum operand, bit 1 mask_src) -> tile configuration id (0 64x64, 1 128x32, 2 128x64 in 4x1
 * waves, 3 128x128, 4 128x64 in 2x2 waves, 5 64x128; < 0 erases).  Replaces what cuDNN's / MIOpen's algorithm search
 * does behind torch.backends.cudnn.benchmark for the reference (mmdet/apis/train.py: cudnn_benchmark of the configs):
 * htd_amd/tuning.py loads the table measured on MI355X by tools/tune_conv_tiles.py; problems not in the table are
 * scored by a model of tile efficiency x wave quantisation.  htd_conv2d_tile_query: the id a launch would use now. */
/* Arithmetic of htd_conv2d_fwd / htd_conv2d_bwd_data / htd_bgemm_nt products: 1 (default) = fp32 through exact three-way
 * bf16 splits of both operands, six v_mfma_f32_32x32x16_bf16 per 16 k with fp32 accumulation (error class of fp32);
 * 0 = v_mfma_f32_32x32x2_f32.  Returns the previous mode (any other argument only queries). */
int htd_conv2d_set_math(int mode);
int htd_conv2d_tile_table_set(int64_t M, int Co, int Ci, int taps, int epi, int cfg);
int htd_conv2d_tile_table_clear(void);
int htd_conv2d_tile_query(int64_t M, int Co, int Ci, int taps, int epi);
int htd_conv2d_fwd(const float *x, const float *w, const float *bias, const float *residual, int res_h,
                   int res_w, float *y, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride,
                   int pad, int dil, int relu, void *workspace, void *stream);
/* Batched NT GEMM on the same MFMA kernel: c[g] = a[g] @ b[g]^T; a [G][M][K], b [G][N][K], c [G][M][N];
 * K % 8 == 0, M % 128 == 0 when G > 1.  Carries PGraph's adjacency x feature contractions
 * (torch.mm calls of htd_bbox_head.py:210,213,214,216 batched over all (image, level) groups). */
int htd_bgemm_nt(const float *a, const float *b, float *c, int G, int M, int N, int K, void *stream);
/* The same over zero-padded groups: counts [G] (DEVICE, int64) says how many leading entries of group g are real; limit
 * bits: 1 = rows of a / c, 2 = rows of b (= columns of c), 4 = the reduction index.  Operands must be zero beyond the count
 * (not read); c is written as zeros there.  No host read of the group sizes (the reference's per-group loop reads them,
 * htd_bbox_head.py:198-219). */
int htd_bgemm_nt_counts(const float *a, const float *b, float *c, int G, int M, int N, int K, const int64_t *counts, int limit,
                        void *stream);
int htd_conv2d_flip_weights(const float *w, float *wT, int Co, int kh, int kw, int Ci, void *stream);
/* Skinny heads (RPN 3+12 channels, fc_cls 81, fc_reg 4): the data gradient reduces over Co, which the MFMA kernel wants
 * as a multiple of 8.  One launch each instead of ATen pad / copy chains: wT rows zero-padded to Co_padded, and
 * y [rows][C_padded] = x [rows][C] followed by zeros for the gradient map. */
int htd_conv2d_flip_weights_padded(const float *w, float *wT, int Co, int Co_padded, int kh, int kw, int Ci, void *stream);
int htd_pad_channels(const float *x, float *y, int64_t rows, int C, int C_padded, void *stream);
/* gx = relu_mask(dgrad(gy) + accum): accum (may be NULL, stride 1 only) is another gradient of the same tensor (the
 * identity branch of a residual block, basic_block/bottleneck `out += identity` resnet.py:278-282), mask_src (may be
 * NULL) the activation whose ReLU produced the conv input (gx is zeroed where mask_src <= 0). */
int htd_conv2d_bwd_data(const float *gy, const float *wT, const float *mask_src, const float *accum, float *gx,
                        int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                        void *workspace, void *stream);
/* htd_conv2d_fwd / htd_conv2d_bwd_data that also leave the largest magnitude of what they store in *amax_out (device scalar, zero
 * or an earlier maximum on entry): the `amax` of the H2 launches (htd_conv2d_fwd_x3h ...) that read the output next. */
int htd_conv2d_fwd_amax(const float *x, const float *w, const float *bias, const float *residual, int res_h, int res_w,
                        float *y, float *amax_out, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                        int dil, int relu, void *workspace, void *stream);
int htd_conv2d_bwd_data_amax(const float *gy, const float *wT, const float *mask_src, const float *accum, float *gx,
                             float *amax_out, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                             int dil, void *workspace, void *stream);
int64_t htd_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Ci, int Co, int kh, int kw,
                                         int stride, int pad, int dil);
int htd_conv2d_bwd_weight(const float *x, const float *gy, float *gw, float *gbias, int B, int H, int W,
                          int Ci, int Co, int kh, int kw, int stride, int pad, int dil, void *workspace,
                          void *stream);
/* The same with gw += and gbias +=: the gradient of a parameter shared by several layers (the RPN convolutions over five
 * pyramid levels, dense_heads/anchor_head.py:123-140; stage 1's classifier reused by stage 2, bbox_heads/htd_bbox_head.py:158)
 * collects in place, call after call on one stream (a fixed summation order). */
int htd_conv2d_bwd_weight_acc(const float *x, const float *gy, float *gw, float *gbias, int B, int H, int W, int Ci,
                              int Co, int kh, int kw, int stride, int pad, int dil, void *workspace, void *stream);
int htd_bias_grad_relu_mask(const float *g, const float *y, float *gm, float *gbias, int64_t rows,
                            int C, void *workspace, void *stream);
/* The same, and max |gm| (max |g| when y is NULL) is left in *amax_out, a device scalar holding zero or an earlier maximum on
 * entry: the `amax` of the H2 data- / weight-gradient launches that read the masked gradient (htd_conv2d_bwd_data_x3h). */
int htd_bias_grad_relu_mask_amax(const float *g, const float *y, float *gm, float *gbias, int64_t rows, int C,
                                 void *workspace, float *amax_out, void *stream);
/* The ResNet stem (backbones/resnet.py:596-607 `conv1`: 7x7, stride 2, padding 3, 3 -> 64 channels; what cuDNN's
 * small-channel first-layer algorithm is to the reference): x [B][H][W][4] fp32 NHWC with the fourth channel zero,
 * w [64][7][7][4] fp32 KRSC, bias [64] or NULL, y [B][Ho][Wo][64], Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1; relu != 0 applies
 * max(., 0).  The reduction runs over filter rows of 7 pixels x 4 channels (224 k per output pixel instead of the 784 the
 * generic kernel spends on an 8-channel image), same split-bf16 arithmetic as the other fp32 convolutions
 * (csrc/conv_stem.hip).  workspace >= htd_conv2d_stem7_workspace_bytes() bytes, 16-byte aligned like the tensors. */
int64_t htd_conv2d_stem7_workspace_bytes(void);
int htd_conv2d_stem7_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int relu,
                         void *workspace, void *stream);
/* The same convolutions with the weight operand split into its three bf16 planes ONCE PER OPTIMIZER STEP instead of once per
 * tile and filter tap inside the kernel (csrc/conv_x3.hip; the role cuDNN's pre-transformed filters play behind
 * backbones/resnet.py:260-300, necks/fpn.py:190-192, dense_heads/rpn_head.py:37-43, htd_bbox_head.py:77-113).
 *   htd_conv2d_x3_planes: w [Co][kh][kw][Ci] -> planes [kh*kw][K/16][3 planes x 2 halves][N rounded up to 128][8 bf16];
 *       transposed = 0: the forward operand (N = Co, K = Ci); transposed = 1: the data-gradient operand (N = Ci, K = Co,
 *       taps reversed) straight from w -- no htd_conv2d_flip_weights image is needed.  `planes` holds
 *       htd_conv2d_x3_planes_bytes(...) bytes (caller-owned, 1.5 x the fp32 weights).
 *   htd_conv2d_x3p_supported: 1 when the two entry points below take the layer: reduction channels % 16 == 0, more than 32
 *       output channels, and either 1x1 (pad 0, any stride) or 3x3 with stride 1, pad 1; the split-bf16 arithmetic must be
 *       selected (htd_conv2d_set_math(1), the default).  Other layers keep htd_conv2d_fwd / htd_conv2d_bwd_data.
 *   htd_conv2d_fwd_x3p / htd_conv2d_bwd_data_x3p: the semantics (epilogue, residual up-sampling, mask_src, accum, split-K
 *       workspace of htd_conv2d_x3p_workspace_bytes) of htd_conv2d_fwd / htd_conv2d_bwd_data; bwd_data is stride 1 only.
 *       Same arithmetic as the default mode of those (six bf16 MFMAs on exact three-way splits, fp32 accumulation). */
int htd_conv2d_x3p_supported(int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
/* Tile table of conv_x3p_kernel, as htd_conv2d_tile_table_set for conv_igemm_kernel: problem (M, Co, Ci, taps, epi) ->
 * configuration id (0 64x64, 1 128x128, 2 128x64, 3 64x128; < 0 erases); epi bit 0: residual / accum operand, bit 1: mask_src,
 * bit 2: the launch runs on the H2 arithmetic (its own entries).  htd_amd/tuning loads the table measured on
 * MI355X by tools/tune_conv_tiles.py --kernel x3p (cudnn_benchmark's role, mmdet/apis/train.py). */
int htd_conv2d_x3p_tile_table_set(int64_t M, int Co, int Ci, int taps, int epi, int cfg);
int htd_conv2d_x3p_tile_table_clear(void);
int htd_conv2d_x3p_tile_query(int64_t M, int Co, int Ci, int taps, int epi);
/* The work decomposition htd_conv2d_fwd_x3p / _bwd_data_x3p use for this problem with tile configuration cfg: layers
 * whose tiles fit on the chip at once are cut along K so that every CU carries the same load (conv_x3.hip, plan_x3p).
 * out[8] = {tiles_a, splits_a, steps_a, splits_b, steps_b, first output row of region B, workgroups, partial floats}. */
int htd_conv2d_x3p_plan_query(int cfg, int64_t M, int Co, int Ci, int kh, int kw, int64_t *out);
int64_t htd_conv2d_x3_planes_bytes(int Co, int kh, int kw, int Ci, int transposed);
int htd_conv2d_x3_planes(const float *w, void *planes, int Co, int kh, int kw, int Ci, int transposed, void *stream);
/* The plane images of many weights in one launch.  desc: DEVICE array of n entries
 *   { const float *w; void *planes; int32 Co, taps, Ci, transposed; int64 block0; }   (40 bytes each)
 * block0 = prefix sum of ceil(taps * (K / 16) * Np / 256) over the entries (Np = N rounded up to 128), total_blocks its end. */
int htd_conv2d_x3_planes_many(const void *desc, int n, int64_t total_blocks, void *stream);
int64_t htd_conv2d_x3p_workspace_bytes(int64_t M, int Co, int Ci, int kh, int kw);
int htd_conv2d_fwd_x3p(const float *x, const void *wplanes, const float *bias, const float *residual, int res_h,
                       int res_w, float *y, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                       int relu, void *workspace, void *stream);
int htd_conv2d_bwd_data_x3p(const float *gy, const void *wplanesT, const float *mask_src, const float *accum,
                            float *gx, int B, int H, int W, int Ci, int Co, int kh, int kw, int pad, void *workspace,
                            void *stream);
/* Round 4: ACTIVATION planes.  A 1x1 / stride-1 layer splits every input element Co / 128 times inside its K loop (8 x for
 * layer3's conv3, backbones/resnet.py:260-300 -- cuDNN's role there); the layer that PRODUCES the map can write its three bf16
 * planes once, next to the fp32 values, from the epilogue that already holds them.
 *   layout: [C/16][3 planes x 2 halves][rows][8 bf16], rows = htd_act_planes_rows(M) = M rounded up to 128 (M = B*H*W pixels),
 *       htd_act_planes_bytes(M, C) bytes, caller-owned (1.5 x the fp32 map); rows >= M are never written and never matter.
 *   htd_act_planes: the planes of an fp32 NHWC map as a pass of its own (maps whose producer is not one of these kernels).
 *   htd_conv2d_fwd_x3q / htd_conv2d_bwd_data_x3q = htd_conv2d_fwd_x3p / htd_conv2d_bwd_data_x3p plus two optional operands:
 *       xplanes / gyplanes != NULL (1x1, stride 1 only): the input map's planes; the fp32 input is then not read and may be
 *           NULL -- both operands reach LDS by DMA, no split in the loop (conv_x3q_kernel); same products in the same order,
 *           so the result equals the fp32-input call bit for bit;
 *       yplanes / gxplanes != NULL (output channels % 16 == 0): the planes of the values stored to y / gx. */
int64_t htd_act_planes_rows(int64_t M);
int64_t htd_act_planes_bytes(int64_t M, int C);
int htd_act_planes(const float *x, void *planes, int64_t M, int C, void *stream);
int htd_conv2d_fwd_x3q(const float *x, const void *xplanes, const void *wplanes, const float *bias, const float *residual,
                       int res_h, int res_w, float *y, void *yplanes, float *amax_out, int B, int H, int W, int Ci, int Co, int kh, int kw,
                       int stride, int pad, int relu, void *workspace, void *stream);
int htd_conv2d_bwd_data_x3q(const float *gy, const void *gyplanes, const void *wplanesT, const float *mask_src,
                            const float *accum, float *gx, void *gxplanes, float *amax_out, int B, int H, int W, int Ci, int Co, int kh, int kw,
                            int pad, void *workspace, void *stream);

/* "H2" arithmetic for the 3x3 layers (round 4): every fp32 product through TWO fp16 pieces per operand -- a0 b0 + a0 b1 + a1 b0,
 * three matrix instructions per block instead of the six of the three-piece bf16 form, operands block-scaled by powers of two
 * (activations: one scale per tensor from its largest magnitude; weights: one per output channel).  Error per product
 * <= 2^-21 |a b| (2^-23.5 rms), like the bf16 form's; role in the reference: cuDNN behind backbones/resnet.py:260-300 and
 * necks/fpn.py, dense_heads/rpn_head.py:25-27.
 *   htd_absmax               *amax = max(*amax, max |x[i]|); *amax is zero (or an earlier maximum) on entry; device scalar
 *   htd_conv2d_x3h_planes    the weight image (htd_conv2d_x3_planes_bytes bytes), transposed = 1: data-gradient operand
 *   htd_conv2d_x3h_planes_many  desc: { const float *w; void *planes; int Co, taps, Ci, transposed; int64_t block0, row0; }
 *   htd_conv2d_fwd_x3h / htd_conv2d_bwd_data_x3h   = htd_conv2d_fwd_x3q / htd_conv2d_bwd_data_x3q with `amax` of the input
 *     amax_out (also on the x3q entry points): the epilogue leaves max |y| of what it stores in this device scalar (zero or
 *     an earlier maximum on entry) -- the consumer's `amax` without a pass over y.  An `amax` that is NOT the tensor's maximum
 *     (the caller's bug) overflows fp16: the output then holds infinities and amax_out says so
 *   htd_conv2d_set_h2        0 / 1 switches the arithmetic off / on (-1: query), returns the previous setting */
int htd_conv2d_set_h2(int on);
/* the weight gradient on the same arithmetic: amax_x / amax_g = max |x| / max |gy| of the two tensors (device scalars);
 * accumulate != 0: htd_conv2d_bwd_weight_acc's semantics; workspace: htd_conv2d_wgrad_workspace_bytes */
int htd_conv2d_bwd_weight_h2_supported(int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
int htd_conv2d_bwd_weight_h2(const float *x, const float *gy, const float *amax_x, const float *amax_g, float *gw,
                             float *gbias, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                             int dil, int accumulate, void *workspace, void *stream);
int htd_conv2d_x3h_supported(int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
/* 1 when htd_conv2d_fwd_x3h also takes this strided layer (3x3, stride 2, pad 1 -- `conv2` of a ResNet stage's first block,
 * backbones/resnet.py:260-300): nine taps on the kernel's 1x1 loop.  Workspace: htd_conv2d_x3p_workspace_bytes(M, Co, Ci, 9, 1). */
int htd_conv2d_x3h_strided_supported(int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
/* Strided data gradients on the H2 kernel: one launch per parity class of gx's pixels with that class's tap list (what
 * htd_conv2d_bwd_data does on conv_igemm_kernel); 3x3 / stride 2 / pad 1 and 1x1 / stride 2 / pad 0 (`conv2` and `downsample` of a
 * ResNet stage's first block, backbones/resnet.py:260-300).  gy [B][Ho][Wo][Co], amax = device scalar with max |gy|, wplanesT =
 * htd_conv2d_x3h_planes(transposed = 1) of the layer's weights, mask_src / amax_out may be NULL. */
int htd_conv2d_bwd_data_x3h_strided_supported(int Ci, int Co, int kh, int kw, int stride, int pad, int dil);
int64_t htd_conv2d_bwd_data_x3h_strided_workspace_bytes(int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad);
int htd_conv2d_bwd_data_x3h_strided(const float *gy, const float *amax, const void *wplanesT, const float *mask_src, float *gx,
                                    float *amax_out, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                                    void *workspace, void *stream);
int htd_absmax(const float *x, int64_t n, float *amax, void *stream);
int htd_conv2d_x3h_planes(const float *w, void *planes, int Co, int kh, int kw, int Ci, int transposed, void *stream);
int htd_conv2d_x3h_planes_many(const void *desc, int n, int64_t total_blocks, int64_t total_rows, void *stream);
int htd_conv2d_fwd_x3h(const float *x, const float *amax, const void *wplanes, const float *bias, const float *residual,
                       int res_h, int res_w, float *y, void *yplanes, float *amax_out, int B, int H, int W, int Ci,
                       int Co, int kh, int kw, int stride, int pad, int relu, void *workspace, void *stream);
int htd_conv2d_bwd_data_x3h(const float *gy, const float *amax, const void *wplanesT, const float *mask_src,
                            const float *accum, float *gx, void *gxplanes, float *amax_out, int B, int H, int W,
                            int Ci, int Co, int kh, int kw, int pad, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------
 * Deformable convolution v1 / v2 (mask == NULL => v1 = the 'DCN' the HTD config uses,
 * configs/htd/htd_resnet101_dcn_2x_mstrain.py:142; built at backbones/resnet.py:186-194).  The reference-era
 * native entry points deform_conv_forward / _backward_input / _backward_parameters
 * (build/lib/mmdet/ops/dcn/deform_conv.py:51-56,75-81,85-91; modulated :144-167) are im2col + GEMM; here the
 * gather/scatter halves are exported and the GEMM halves are htd_conv2d_fwd / _bwd_data / _bwd_weight (1x1)
 * on the column matrix, so the contraction runs on the same MFMA kernels as every other convolution:
 *   fwd:  columns = deform_im2col(x, offset, mask); y = conv2d_fwd(columns as [M][1][1][K], w[Co][K])
 *   bwd:  gcol = conv2d_bwd_data(gy, w); (gx, goffset, gmask) = deform_col2im(gcol); gw = conv2d_bwd_weight
 *   x [B][H][W][C]   offset [B][Ho][Wo][dg*kh*kw*2] (dy,dx pairs)   mask [B][Ho][Wo][dg*kh*kw] or NULL
 *   columns / gcolumns [B*Ho*Wo][kh*kw*C] (tap-major: same k order as KRSC weights)
 *   gx must be zero-initialised (atomic accumulation); gx / goffset / gmask may be NULL to skip them.
 * ---------------------------------------------------------------------------------- */
int64_t htd_deform_columns_bytes(int B, int H, int W, int C, int kh, int kw, int stride, int pad, int dil);
int htd_deform_im2col(const float *x, const float *offset, const float *mask, float *columns, int B,
                      int H, int W, int C, int kh, int kw, int stride, int pad, int dil,
                      int deform_groups, void *stream);
int htd_deform_col2im(const float *x, const float *offset, const float *mask, const float *gcolumns,
                      float *gx, float *goffset, float *gmask, int B, int H, int W, int C, int kh,
                      int kw, int stride, int pad, int dil, int deform_groups, void *stream);
/* bf16 forms for the mixed-precision mode (BASELINE configs[3]): x, columns and gradient columns in bf16; offsets,
 * masks, every interpolation and accumulation, gx / goffset / gmask in fp32. */
int htd_deform_im2col_bf16(const void *x, const float *offset, const float *mask, void *columns, int B, int H, int W,
                           int C, int kh, int kw, int stride, int pad, int dil, int deform_groups, void *stream);
int htd_deform_col2im_bf16(const void *x, const float *offset, const float *mask, const void *gcolumns, float *gx,
                           float *goffset, float *gmask, int B, int H, int W, int C, int kh, int kw, int stride, int pad,
                           int dil, int deform_groups, void *stream);

/* ------------------------------------------------------------------------------------
 * SFA global pooling (GlobalContextHead.forward global_context_head.py:386,
 * nn.AdaptiveAvgPool2d(1)) and the 7x7 AvgPool of the reg branch (htd_bbox_head.py:122,188):
 *   out[b][c] = mean over P positions of x[b][p][c];  bwd: gx[b][p][c] = g[b][c] / P.
 * ---------------------------------------------------------------------------------- */
int htd_global_avg_pool_fwd(const float *x, float *out, int64_t B, int P, int C, void *stream);
int htd_global_avg_pool_bwd(const float *g, float *gx, int64_t B, int P, int C, void *stream);
/* gx += g / P: the pooled map has a second consumer whose gradient gx already holds (adaptative_roi_extractor.py:72-86: a
 * level's RoI features feed the attention pooling and the weighted sum). */
int htd_global_avg_pool_bwd_acc(const float *g, float *gx, int64_t B, int P, int C, void *stream);

/* ------------------------------------------------------------------------------------
 * GroupNorm + ReLU on NHWC RoI tiles (GN36 over 576 channels, htd_bbox_head.py:48,89,111):
 *   x [n][P][C], G groups; saves mean/rstd [n][G] for backward.
 * ---------------------------------------------------------------------------------- */
int htd_group_norm_relu_fwd(const float *x, const float *gamma, const float *beta, float *y,
                            float *mean, float *rstd, int64_t n, int P, int C, int G, float eps,
                            int relu, void *stream);
int htd_group_norm_relu_bwd(const float *x, const float *y, const float *gamma, const float *mean,
                            const float *rstd, const float *gy, float *gx, float *ggamma,
                            float *gbeta, int64_t n, int P, int C, int G, int relu, void *stream);
/* Bit-reproducible forms of the two backward passes that summed with float atomics (their run-to-run rounding differences
 * reached every parameter upstream of P6 and made two ranks' replicas drift apart in the last bit): per-tile sums go through a
 * caller-owned workspace and are added in a fixed order.  htd_group_norm_relu_bwd_ws: workspace 2 * n * C floats, ggamma /
 * gbeta overwritten.  htd_fuse_global_bwd_global_ws: workspace (n + ceil(n / 64) * B) * C floats, grad_global [B][C]
 * overwritten, C % 4 == 0. */
int htd_group_norm_relu_bwd_ws(const float *x, const float *y, const float *gamma, const float *mean, const float *rstd,
                               const float *gy, float *gx, float *ggamma, float *gbeta, int64_t n, int P, int C, int G,
                               int relu, void *workspace, void *stream);
/* Round 4: the same two passes leaving the largest magnitude of what they store in a device scalar (zero or an earlier maximum on
 * entry) -- the `amax` operand of the H2 convolution that consumes the tensor (htd_conv2d_fwd_x3h / htd_conv2d_bwd_data_x3h /
 * htd_conv2d_bwd_weight_h2), without a pass of htd_absmax.  The backward leaves it on the shapes its bandwidth-form kernel takes
 * (htd_group_norm_bwd_amax_supported; the 7x7 x 576-channel tiles of htd_bbox_head.py:77-113 are one). */
int htd_group_norm_relu_fwd_amax(const float *x, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                                 int64_t n, int P, int C, int G, float eps, int relu, float *amax_out, void *stream);
int htd_group_norm_bwd_amax_supported(int P, int C, int G);
int htd_group_norm_relu_bwd_amax(const float *x, const float *y, const float *gamma, const float *mean, const float *rstd,
                                 const float *gy, float *gx, float *ggamma, float *gbeta, int64_t n, int P, int C, int G,
                                 int relu, void *workspace, float *amax_out, void *stream);
int htd_fuse_global_bwd_global_ws(const float *grad, const float *rois, float *grad_global, int64_t n, int P, int C, int B,
                                  void *workspace, void *stream);

/* ------------------------------------------------------------------------------------
 * Frozen-statistics BatchNorm (norm_eval=True, backbones/resnet.py:640-649) folded into the preceding
 * convolution: w'[co][k] = w[co][k]*s, b'[co] = beta - mean*s, s = gamma/sqrt(var+eps), k over kh*kw*Ci (KRSC
 * row).  bwd maps the gradients of the folded tensors back: gw = gw'*s, gbeta = gb',
 * ggamma = (<gw'[co], w[co]> - mean*gb') / sqrt(var+eps) -- exactly d/dgamma, d/dbeta, d/dw of conv -> BN(eval).
 * ---------------------------------------------------------------------------------- */
int htd_bn_fold_fwd(const float *w, const float *gamma, const float *beta, const float *mean,
                    const float *var, float eps, float *w_folded, float *b_folded, int Co, int K,
                    void *stream);
int htd_bn_fold_bwd(const float *w, const float *gamma, const float *mean, const float *var, float eps,
                    const float *gw_folded, const float *gb_folded, float *gw, float *ggamma,
                    float *gbeta, int Co, int K, void *stream);
/* The same two operations for MANY conv + BN pairs in one launch each (a whole ResNet stage: up to 70 pairs), driven by
 * a device table.  Forward table entry (80 bytes): { const float *w, *gamma, *beta, *mean, *var; float *wf, *bf, *wT;
 * int Co, Ci, taps, tile0; } -- wT (may be NULL) also receives the flipped / transposed image of the FOLDED weights that
 * htd_conv2d_bwd_data takes (htd_conv2d_flip_weights); tile0 = prefix sum of taps * ceil(Co/32) * ceil(Ci/32).  An entry with
 * gamma == NULL has no BN: only wT is written, from w itself (every flip of a step's plain convolutions in one launch).
 * Backward table entry (88 bytes): { const float *w, *gamma, *mean, *var, *gwf, *gbf; float *gw, *ggamma, *gbeta;
 * int Co, K, row0, pad; } -- row0 = prefix sum of Co, K = taps * Ci (K % 4 == 0). */
int htd_bn_fold_many_fwd(const void *desc, int n_layers, int total_tiles, float eps, void *stream);
int htd_bn_fold_many_bwd(const void *desc, int n_layers, int total_rows, float eps, void *stream);

/* ------------------------------------------------------------------------------------
 * MaxIoUAssigner for a whole batch (replaces MaxIoUAssigner.assign / assign_wrt_overlaps,
 * mmdet/core/bbox/assigners/max_iou_assigner.py:60-212, and its bbox_overlaps call,
 * iou_calculators/iou2d_calculator.py:43-158, run per image on the k x A IoU matrix).
 *   boxes [B][A][4] (box_shared = 0) or [A][4] shared by every image (RPN anchors); box_valid [B][A];
 *   gts [B][K][4] zero padded, gt_valid [B][K].
 *   assigned [B][A] int64: -1 ignore / invalid box, 0 negative, k+1 matched to gt k;  max_overlaps [B][A].
 *   match_low_quality follows gt_max_assign_all = True (:187-199); it needs a workspace of
 *   htd_max_iou_assign_workspace_bytes(B, A, K) bytes.
 * IoU arithmetic is the reference's fp32 expression evaluated without FMA contraction: assignments are bit-exact.
 * ---------------------------------------------------------------------------------- */
int64_t htd_max_iou_assign_workspace_bytes(int B, int A, int K);
int htd_max_iou_assign(const float *boxes, int box_shared, const uint8_t *box_valid, const float *gts,
                       const uint8_t *gt_valid, int B, int A, int K, float pos_iou_thr, float neg_iou_thr,
                       float min_pos_iou, int match_low_quality, int64_t *assigned, float *max_overlaps,
                       void *workspace, void *stream);

/* BBoxHead.get_targets / _get_target_single (bbox_heads/bbox_head.py:85-146) on fixed sample slots: labels
 * (background = num_classes unless the slot is a positive), label weights (1 on used slots), bbox2delta targets
 * (delta_xywh_bbox_coder.py:78-120; means4 / stds4 are HOST arrays) and their weights (1 on positives).  n rows. */
int htd_roi_targets(const float *boxes, const float *gt_boxes, const int64_t *gt_labels, const uint8_t *is_pos,
                    const uint8_t *valid, int64_t n, int num_classes, const float *means4, const float *stds4,
                    int64_t *labels, float *label_weights, float *bbox_targets, float *bbox_weights, void *stream);
/* delta2bbox (delta_xywh_bbox_coder.py:123-204) for 4-column deltas + clip to the image of each row
 * (row / rows_per_img -> lim_wh [img][2] = (w, h) on the device; NULL = no clip) + rows with keep == 0 zeroed
 * (keep may be NULL): RPN proposal decode (rpn_head.py:122-140) and BBoxHead.refine_bboxes (bbox_head.py:227-304). */
int htd_delta2bbox_clip(const float *rois, const float *deltas, const float *lim_wh, const uint8_t *keep, int64_t n,
                        int64_t rows_per_img, const float *means4, const float *stds4, float wh_ratio_clip,
                        float *out, void *stream);
/* map_roi_levels (roi_extractors/single_level_roi_extractor.py:32-51, bbox_heads/htd_bbox_head.py:129-135):
 * lvls[i] = clamp(floor(log2(sqrt(w_i * h_i) / finest_scale + 1e-6)), 0, num_levels - 1) for rois (n, 5), int64 out. */
int htd_map_roi_levels(const float *rois, int64_t *lvls, int64_t n, int num_levels, float finest_scale, void *stream);
/* The RPN head outputs of all pyramid levels <-> the flat per-anchor tensors that targets, loss and proposal generation use
 * (dense_heads/anchor_head.py:172-269, rpn_head.py:78-168: permute(0, 2, 3, 1).reshape per level, then concatenate).
 * y[l]: level l's merged head output [B][pix[l]][C], channel a < na = objectness of anchor a, na + 4 a + j = delta j of
 * anchor a, the rest padding.  gather: cls [B][A] and reg [B][A][4], A = na * sum(pix), level-major; scatter: the transpose
 * (gradient of y[l], padding channels zero).  One launch each for all levels. */
int htd_rpn_heads_gather(const float *const *y, const int64_t *pix, int L, int B, int C, int na, float *cls, float *reg,
                         void *stream);
int htd_rpn_heads_scatter(const float *gcls, const float *greg, float *const *gy, const int64_t *pix, int L, int B, int C,
                          int na, void *stream);
/* The fixed-slot sampling result of a batch (SamplingResult, core/bbox/samplers/sampling_result.py:40-60, for every image of
 * htd_roi_head.py:254-264,292-310) from htd_random_sample's slot order and counts: boxes (B,S,4) zeroed past the drawn
 * count, valid / is_pos / pos_is_gt (B,S) as bytes, the assigned gt's box and label per slot.  Candidates are
 * [gts (K, when add_gt) | proposals (P) | padding]; assigned (B,A) holds 1-based gt indices. */
int htd_static_samples_finish(const float *gts, const uint8_t *gvalid, const int64_t *glabels, const float *props,
                              const int64_t *assigned, const int64_t *order, const int64_t *counts, int B, int K, int P,
                              int A, int S, int add_gt, float *boxes, uint8_t *valid, uint8_t *is_pos,
                              float *pos_gt_boxes, int64_t *pos_gt_labels, uint8_t *pos_is_gt, void *stream);

/* RPN loss of the whole batch in one pass (AnchorHead.loss / loss_single, dense_heads/anchor_head.py:373-488, with the
 * targets of _get_targets_single :172-269 and bbox2delta formed on the fly).  cls [B*A] logits (one sigmoid channel),
 * reg [B*A][4], anchors [A][4], gts [B][K][4], assigned [B][A] (MaxIoUAssigner output), pos / neg [B][A] sample masks.
 * partial [htd_rpn_loss_partial_rows()][2] = per-block (sum BCE, sum SmoothL1) -- reduce over rows in order;
 * grad_cls [B*A], grad_reg [B*A][4] = derivatives of those sums (scale by 1/avg_factor * loss_weight). */
/* Both losses of a RoI head stage in one pass (BBoxHead.loss, bbox_heads/bbox_head.py:148-186: CrossEntropyLoss over the
 * class logits with per-row weights, SmoothL1Loss over the class-agnostic box deltas of the foreground rows, accuracy):
 * per-block partial sums partial[htd_roi_head_loss_partial_rows()][4] = {sum w*CE, #(w > 0), sum bw*SmoothL1, #correct} in a
 * fixed grid (the caller adds them in row order: reproducible) and the derivatives of the two sums, grad_cls [n][NC] and
 * grad_box [n][4].  NC <= 128; bbox_pred == NULL: classification part only. */
int htd_roi_head_loss_partial_rows(void);
int htd_roi_head_loss(const float *cls_score, const int64_t *labels, const float *label_weights, const float *bbox_pred,
                      const float *bbox_targets, const float *bbox_weights, int64_t n, int NC, int num_fg, float beta,
                      float *partial, float *grad_cls, float *grad_box, void *stream);
int htd_rpn_loss_partial_rows(void);
int htd_rpn_loss(const float *cls, const float *reg, const float *anchors, const float *gts, const int64_t *assigned,
                 const uint8_t *pos, const uint8_t *neg, int B, int A, int K, const float *means4, const float *stds4,
                 float beta, float pos_weight, float *partial, float *grad_cls, float *grad_reg, void *stream);

/* nn.MaxPool2d(kernel, stride, padding) of the ResNet stem (backbones/resnet.py:509,629) on NHWC maps
 * x [B][H][W][C] -> y [B][Ho][Wo][C], C % 4 == 0, padding = -inf, floor mode.  idx (may be NULL for inference; int32
 * [B][Ho][Wo][C]) records the input pixel hi*W+wi of the first maximum of each window; bwd sends the gradient there
 * (gather form, no atomics). */
/* Gradient of the nearest-neighbour up-sampling of the FPN top-down path (necks/fpn.py:177-186; its forward lives in the
 * lateral convolution's epilogue, htd_conv2d_fwd res_h / res_w): out [B][h][w][C] = sum of g [B][H][W][C] over the fine pixels
 * whose source pixel it is under ATen's rule min(floor(dst * in / out), in - 1).  C % 4 == 0. */
int htd_upsample_nearest_bwd(const float *g, float *out, int B, int H, int W, int h, int w, int C, void *stream);
int htd_max_pool2d_fwd(const float *x, float *y, int *idx, int B, int H, int W, int C, int k, int stride, int pad,
                       void *stream);
/* The same, also leaving max |y| in *amax_out (device scalar, zero or an earlier maximum on entry) for the H2 launches that read
 * the pooled map (htd_conv2d_fwd_x3h). */
int htd_max_pool2d_fwd_amax(const float *x, float *y, int *idx, int B, int H, int W, int C, int k, int stride, int pad,
                            float *amax_out, void *stream);
int htd_max_pool2d_bwd(const float *g, const int *idx, float *gx, int B, int H, int W, int C, int k, int stride,
                       int pad, void *stream);

/* bf16 forward convolution (fp32 accumulate) on v_mfma_f32_32x32x16_bf16: groundwork for the bf16 configurations
 * (BASELINE configs[2], [3]); same semantics as htd_conv2d_fwd with x, w, residual, y in bf16 (NHWC / KRSC) and bias in
 * fp32.  Ci % 32 == 0, Co % 4 == 0. */
int htd_conv2d_fwd_bf16(const void *x, const void *w, const float *bias, const void *residual, void *y, int B, int H,
                        int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil, int relu, void *stream);
/* htd_conv2d_fwd_bf16 with the residual [B][res_h][res_w][Co] read through nearest up-sampling to the output size
 * (res_h = 0: a same-size residual): the FPN top-down sum inside the lateral convolution, mmdet necks/fpn.py:177-186
 * (F.interpolate(..., mode='nearest') + add), for bf16 maps. */
int htd_conv2d_fwd_bf16_up(const void *x, const void *w, const float *bias, const void *residual, int res_h, int res_w,
                           void *y, int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dil,
                           int relu, void *stream);
/* htd_upsample_nearest_bwd on bf16 maps (fp32 sums, one rounding). */
int htd_upsample_nearest_bwd_bf16(const void *g, void *out, int B, int H, int W, int h, int w, int C, void *stream);
/* The data-gradient form of the bf16 convolution: gx = (conv(gy, wT) + accum) * (mask_src > 0), every map bf16;
 * (H, W) are gy's; stride-1 layers only; wT = the transposed, tap-flipped weights [Ci'][kh][kw][Co'] of
 * htd_weights_prep_bf16 (Ci = channels of gy, Co = channels of gx); pad = dil*(k-1) - layer padding.  mask_src (the
 * saved input activation: ReLU backward of its producer) and accum (the gradient arriving over the identity branch)
 * may be NULL. */
int htd_conv2d_dgrad_bf16(const void *gy, const void *wT, const void *mask_src, const void *accum, void *gx, int B, int H,
                          int W, int Ci, int Co, int kh, int kw, int pad, int dil, void *stream);
/* One launch per layer and step: fp32 (BN-folded) weights w [Co][kh][kw][Ci] -> wb (bf16, same layout) and
 * wT [Ci][kh][kw][Co] (bf16, taps flipped); either output may be NULL. */
int htd_weights_prep_bf16(const float *w, void *wb, void *wT, int Co, int kh, int kw, int Ci, void *stream);
/* The same for many layers in one launch.  desc: DEVICE array of n entries
 *   { const float *w; void *wb; void *wT (may be 0); int32 Co, taps, Ci, 0; int64 tile0 }   (48 bytes each),
 * tile0 = prefix sum of taps * ceil(Co / 32) * ceil(Ci / 32) over the entries, total_tiles its end. */
int htd_weights_prep_bf16_many(const void *desc, int n, int64_t total_tiles, void *stream);
/* fp32 column sums of a bf16 matrix g [rows][C] (bias gradients), deterministic two-stage; C % 4 == 0. */
int64_t htd_colsum_bf16_workspace_bytes(int64_t rows, int C);
int htd_colsum_bf16(const void *g, float *out, int64_t rows, int C, void *workspace, void *stream);
/* bf16 weight gradient: gw [Co][kh][kw][Ci] fp32 from x, gy in bf16 (NHWC); both K-major operands are consumed through
 * the transposing LDS read ds_read_b64_tr_b16; deterministic split-K through `workspace`
 * (htd_conv2d_wgrad_bf16_workspace_bytes).  Ci % 8 == 0, Co % 8 == 0. */
int64_t htd_conv2d_wgrad_bf16_workspace_bytes(int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                                              int dil);
int htd_conv2d_bwd_weight_bf16(const void *x, const void *gy, float *gw, int B, int H, int W, int Ci, int Co, int kh,
                               int kw, int stride, int pad, int dil, void *workspace, void *stream);
/* ... and gbias [Co] (fp32) = column sums of gy from the same launch (the bias / BN-beta gradient of the layer), as
 * htd_conv2d_bwd_weight does in fp32. */
int htd_conv2d_bwd_weight_bf16_bias(const void *x, const void *gy, float *gw, float *gbias, int B, int H, int W, int Ci, int Co,
                                    int kh, int kw, int stride, int pad, int dil, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------
 * Grouped convolutions of the ResNeXt bottlenecks: conv2 of backbones/resnext.py:33-84 is
 * nn.Conv2d(width, width, 3, groups=64) -- ATen/MIOpen in the reference -- or, with dcn=..., the grouped
 * deformable conv (deform_conv_ext with group=64, build/lib/mmdet/ops/dcn/deform_conv.py:51-58).
 *   x  [B][H][W][C]   w  [C][kh][kw][C/groups] (grouped KRSC)   y [B][Ho][Wo][C]   (in == out channels)
 *   C % 16 == 0; channels per group 4, 8, 16 or a multiple of 16; kh*kw <= 9.
 * The kernels take the weights PACKED into 16x16 channel-slab tiles (block-diagonal across narrow groups):
 * htd_gconv2d_pack_weights(w, wp, ..., transpose) with transpose = 0 for fwd, 1 for bwd_data;
 * htd_gconv2d_packed_floats() floats each.
 *   cols = 1: x (fwd, bwd_weight) / gx (bwd_data) is the gathered column buffer [M][kh*kw][C] of the deformable
 *   conv (htd_deform_im2col) instead of an image; M = B*Ho*Wo.
 * bwd_weight: deterministic split over pixels through `workspace` (htd_gconv2d_wgrad_workspace_bytes).
 * ---------------------------------------------------------------------------------- */
int64_t htd_gconv2d_packed_floats(int C, int groups, int kh, int kw);
int htd_gconv2d_pack_weights(const float *w, float *wp, int C, int groups, int kh, int kw, int transpose, void *stream);
int htd_gconv2d_fwd(const float *x, const float *wp, const float *bias, float *y, int B, int H, int W, int C, int groups,
                    int kh, int kw, int stride, int pad, int dil, int relu, int cols, void *stream);
int htd_gconv2d_bwd_data(const float *gy, const float *wpT, float *gx, int B, int H, int W, int C, int groups, int kh,
                         int kw, int stride, int pad, int dil, int cols, void *stream);
int64_t htd_gconv2d_wgrad_workspace_bytes(int B, int H, int W, int C, int groups, int kh, int kw, int stride, int pad,
                                          int dil);
int htd_gconv2d_bwd_weight(const float *x, const float *gy, float *gw, int B, int H, int W, int C, int groups, int kh,
                           int kw, int stride, int pad, int dil, int cols, void *workspace, void *stream);

/* ------------------------------------------------------------------------------------
 * On-device data pipeline of one batch of decoded 8-bit images: in ONE pass
 *   Resize   mmcv.imrescale -> cv2.resize(INTER_LINEAR) on uint8 (datasets/pipelines/transforms.py:202-231): OpenCV's
 *            fixed-point bilinear (11-bit weights; exact 2x downscale = 2x2 mean), sizes chosen by the host;
 *   flip     mmcv.imflip (transforms.py:440-444): bit 0 horizontal, bit 1 vertical, both = diagonal;
 *   Normalize mmcv.imnormalize (transforms.py:563-575): BGR->RGB if to_rgb, (x - mean) in fp32, * (1/(double)std);
 *   Pad + collate  mmcv.impad_to_multiple (transforms.py:496-505) and mmcv.parallel.collate's pad to the batch
 *            maximum: every pixel outside an image's dst_h x dst_w is pad_val.
 *   src      packed HWC (3 channels) uint8 images, image b starts at byte src_off[b]  (device)
 *   meta     48 B per image (device, 8 B aligned): int32 {src_h, src_w, dst_h, dst_w, flip, area2, 0, 0} then double
 *            {scale_x, scale_y} = 1.0 / ((double)dst / src), the values cv2.resize builds its tables from; area2 = 1
 *            iff src_h == 2*dst_h && src_w == 2*dst_w; the caller guarantees dst_h <= Hp, dst_w <= Wp and that
 *            every image lies inside `src`
 *   out      [B][Hp][Wp][3] fp32, written in full (= torch channels_last of (B,3,Hp,Wp))
 * ---------------------------------------------------------------------------------- */
int htd_image_batch_pipeline(const uint8_t *src, const int64_t *src_off, const int *meta, float *out, int B, int Hp,
                             int Wp, float mean0, float mean1, float mean2, float std0, float std1, float std2,
                             int to_rgb, float pad_val, void *stream);

/* ------------------------------------------------------------------------------------
 * SGD with momentum and weight decay on the flat parameter buffer (the update the mmcv
 * OptimizerHook performs after the DDP all-reduce; configs/_base_/schedules/schedule_1x.py:2):
 *   g = grad*grad_scale + wd*p ; m = momentum*m + g ; p -= lr*m.     lr is a device scalar
 *   so LR warm-up needs no re-capture.
 * ---------------------------------------------------------------------------------- */
int htd_sgd_momentum_step(float *param, const float *grad, float *momentum_buf, int64_t n,
                          const float *lr_dev, float momentum, float weight_decay,
                          float grad_scale, void *stream);

/* ----------------------------------------------------------------------------------
 * PGraph adjacency (HTDBBoxHead.forward, mmdet/models/roi_heads/bbox_heads/htd_bbox_head.py:198-219), batched over the
 * (image, level) groups of a call: group g = rows 0..counts[g]-1 of a [G][npad] padding (npad % 64 == 0, <= 1024).
 *   adjacency    boxes [G][npad][4] -> A_local = D^-1/2 M D^-1/2, M = (bbox_overlaps with unit diagonal) > 0 (:207-210);
 *                dinv_scratch: G * npad floats
 *   softmax_fwd  A_glob = softmax_row((1 - M) * sim) over the valid columns (:211,214-215; local pairs keep logit 0);
 *                M is read back as A_local > 0; rows past counts[g] are zeros
 *   softmax_bwd  gradient of A_glob with respect to sim
 * ---------------------------------------------------------------------------------- */
int htd_pgraph_adjacency(const float *boxes, const int64_t *counts, float *A_local, float *dinv_scratch, int G, int npad,
                         void *stream);
int htd_pgraph_softmax_fwd(const float *sim, const float *A_local, const int64_t *counts, float *A_glob, int G, int npad,
                           void *stream);
int htd_pgraph_softmax_bwd(const float *gA, const float *A_glob, const float *A_local, const int64_t *counts, float *gsim,
                           int G, int npad, void *stream);
/* Group gathers of HTDBBoxHead.forward (`x[mask]` per (image, level), htd_bbox_head.py:198-206) for all groups at once:
 * out[i] = valid[i] ? x[rows[i]] : 0 for the n_out = G * npad padded slots; out [n_out][Fo] (Fo >= F, zero tail) or, with
 * transposed != 0, [G][F][npad] (the K-major operand of A_local @ x).  htd_pgraph_scatter is the adjoint (gx zeroed, then
 * gx[rows[i]] = g[i]; every RoI occupies at most one valid slot). */
int htd_pgraph_gather(const float *x, const int64_t *rows, const unsigned char *valid, float *out, int64_t n_out, int F,
                      int Fo, int G, int transposed, void *stream);
int htd_pgraph_scatter(const float *g, const int64_t *rows, const unsigned char *valid, float *gx, int64_t n_out, int64_t N,
                       int F, int Fo, int G, int transposed, void *stream);

/* Row selection and its adjoint for the stage-2 positives of the RoI tiles (`bbox_feats[pos_inds]`, roi_heads/htd_roi_head.py:163-166):
 * out[i] = x[rows[i]], i < n / gx[rows[i]] += g[i] (distinct rows).  Rows of F floats, F % 4 == 0, rows[i] in [0, N). */
int htd_rows_gather(const float *x, const int64_t *rows, float *out, int64_t n, int64_t N, int F, void *stream);
int htd_rows_add(const float *g, const int64_t *rows, float *gx, int64_t n, int64_t N, int F, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* HTD_AMD_H */
