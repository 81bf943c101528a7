"""HTDRoIHead: the two-stage decoupled RoI head (driver of SFA, stage 1, BA and PGraph).

Reference: roi_heads/htd_roi_head.py:12-476 (+ base_roi_head.py:8-106).  Same registry name, kwargs,
sub-module names (bbox_roi_extractor.{0,1}, bbox_head.{0,1}, glbctx_head), loss keys
(loss_global, s0.loss_cls, s0.acc, s0.loss_bbox, s1.*) and return structures.

Deliberate fixes ('F'): stage-2 positives are taken from EVERY image of the batch
(the reference hard-codes image ids 0 and 1, :158-169,181-182 -- identical for B <= 2); `_fuse_global`
does not round-trip through the host.
"""
import torch
import torch.nn as nn

from .. import mmcv_ops as M
from ..core import bbox2result, bbox2roi
from ..core.misc import arange_cached, const_tensor
from ..registry import HEADS, build_assigner, build_head, build_roi_extractor, build_sampler


POS_BUCKET = max(1, int(__import__('os').environ.get('HTD_POS_BUCKET', '16')))    # stage-2 regression rows: multiples of this


@HEADS.register_module()
class HTDRoIHead(nn.Module):
    def __init__(self, num_stages, stage_loss_weights, with_global=False, bbox_roi_extractor=None, bbox_head=None,
                 mask_roi_extractor=None, mask_head=None, shared_head=None, train_cfg=None, test_cfg=None):
        super().__init__()
        assert bbox_roi_extractor is not None and bbox_head is not None
        assert shared_head is None, 'Shared head is not supported in Cascade RCNN anymore'
        assert mask_head is None and mask_roi_extractor is None, 'HTD configs have no mask branch'
        self.num_stages, self.stage_loss_weights, self.with_global = num_stages, stage_loss_weights, with_global
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_bbox_head(bbox_roi_extractor, bbox_head)
        self.init_assigner_sampler()

    @property
    def with_bbox(self):
        return hasattr(self, 'bbox_head') and self.bbox_head is not None

    @property
    def with_mask(self):
        return False

    @property
    def with_shared_head(self):
        return False

    def init_bbox_head(self, bbox_roi_extractor, bbox_head):
        self.bbox_roi_extractor = nn.ModuleList()
        self.bbox_head = nn.ModuleList()
        if not isinstance(bbox_roi_extractor, list):
            bbox_roi_extractor = [bbox_roi_extractor for _ in range(self.num_stages)]
        if not isinstance(bbox_head, list):
            bbox_head = [bbox_head for _ in range(self.num_stages)]
        assert len(bbox_roi_extractor) == len(bbox_head) == self.num_stages
        for ext, head in zip(bbox_roi_extractor, bbox_head):
            self.bbox_roi_extractor.append(build_roi_extractor(ext))
            self.bbox_head.append(build_head(head))
        if self.with_global:
            self.glbctx_head = build_head(dict(type='GlobalContextHead', num_ins=5, num_convs=4, in_channels=256,
                                               conv_out_channels=256, num_classes=self.bbox_head[0].num_classes + 1,
                                               loss_weight=3.0))

    def init_assigner_sampler(self):
        self.bbox_assigner, self.bbox_sampler = [], []
        if self.train_cfg is not None:
            for idx, rcnn_train_cfg in enumerate(self.train_cfg):
                self.bbox_assigner.append(build_assigner(rcnn_train_cfg.assigner))
                self.current_stage = idx
                self.bbox_sampler.append(build_sampler(rcnn_train_cfg.sampler, context=self))

    def init_weights(self, pretrained=None):
        for i in range(self.num_stages):
            self.bbox_roi_extractor[i].init_weights()
            self.bbox_head[i].init_weights()
        if self.with_global:
            self.glbctx_head.init_weights()

    def _fuse_global(self, roi_feats, global_feat, rois):
        assert roi_feats.size(0) == rois.size(0)
        return M.fuse_global(roi_feats, rois, global_feat)

    # ------------------------------------------------------------------ per-stage forward
    def _bbox_forward(self, stage, x, rois, global_feat=None, sampling_results=None, img_metas=None, taps=None):
        """taps (training): mmcv_ops.PyramidTaps over the pyramid levels -- the three RoIAlign consumers of a step
        (both extractors and BA) then share one gradient map per level instead of summing three."""
        extractor, enhanced_extractor = self.bbox_roi_extractor[0], self.bbox_roi_extractor[1]
        feats = taps if taps is not None else x[:extractor.num_inputs]
        if stage == 0:
            bbox_feats = extractor(feats, rois)
            if self.with_global:
                bbox_feats = self._fuse_global(bbox_feats, global_feat, rois)
            cls_score, bbox_pred = self.bbox_head[0](bbox_feats)
            return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=bbox_feats)
        head = self.bbox_head[stage]
        bbox_feats = extractor(feats, rois)
        if sampling_results:
            # training: BA + reg branch on the positives only; rows of `rois` are [pos_i, neg_i] per image
            pos_rois = bbox2roi([res.pos_bboxes for res in sampling_results])
            pos_rows, start = [], 0
            for res in sampling_results:
                npos = res.pos_bboxes.size(0)
                pos_rows.append(torch.arange(start, start + npos, device=rois.device))
                start += npos + res.neg_bboxes.size(0)
            pos_rows = torch.cat(pos_rows)
            enhanced = enhanced_extractor(feats, pos_rois)
            pos_bbox_feat = torch.index_select(bbox_feats, 0, pos_rows)
            per_img = tuple(r.pos_bboxes.size(0) + r.neg_bboxes.size(0) for r in sampling_results)
            cls_score, bbox_pred = head(bbox_feats, pos_bbox_feat, feats, rois, self.bbox_head[0].fc_cls, enhanced,
                                        pos_rois, global_feat if self.with_global else None, rois_per_img=per_img)
            full = cls_score.new_zeros(cls_score.size(0), 4).index_put((pos_rows, ), bbox_pred)
            return dict(cls_score=cls_score, bbox_pred=full)
        enhanced = enhanced_extractor(feats, rois)
        cls_score, bbox_pred = head(bbox_feats, bbox_feats, feats, rois, self.bbox_head[0].fc_cls, enhanced, rois,
                                    global_feat if self.with_global else None)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred)

    def _bbox_forward_train(self, stage, x, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, img_metas,
                            global_feat=None, taps=None):
        rois = bbox2roi([res.bboxes for res in sampling_results])
        bbox_results = self._bbox_forward(stage, x, rois, global_feat, sampling_results, img_metas, taps)
        bbox_targets = self._targets(stage, sampling_results, rcnn_train_cfg)
        loss_bbox = self.bbox_head[stage].loss(bbox_results['cls_score'], bbox_results['bbox_pred'], rois,
                                               *bbox_targets)
        bbox_results.update(loss_bbox=loss_bbox, rois=rois, bbox_targets=bbox_targets)
        return bbox_results

    def _targets(self, stage, sampling_results, cfg):
        """BBoxHead.get_targets (bbox_head.py:85-139) for the whole batch in a handful of launches: rows are
        [pos_i ; neg_i] per image, positives carry their gt label and encoded deltas, everything has weight 1
        (pos_weight <= 0), negatives the background label."""
        head = self.bbox_head[stage]
        if cfg.pos_weight > 0 or head.reg_decoded_bbox:
            return head.get_targets(sampling_results, None, None, cfg)
        npos = [r.pos_bboxes.size(0) for r in sampling_results]
        nneg = [r.neg_bboxes.size(0) for r in sampling_results]
        N = sum(npos) + sum(nneg)
        dev = sampling_results[0].pos_bboxes.device
        pos_rows, start = [], 0
        for a, b in zip(npos, nneg):
            pos_rows.append(torch.arange(start, start + a, device=dev))
            start += a + b
        pos_rows = torch.cat(pos_rows)
        pos_b = torch.cat([r.pos_bboxes for r in sampling_results])
        labels = pos_b.new_full((N, ), head.num_classes, dtype=torch.long)
        bbox_targets = pos_b.new_zeros(N, 4)
        bbox_weights = pos_b.new_zeros(N, 4)
        if pos_rows.numel():
            labels[pos_rows] = torch.cat([r.pos_gt_labels for r in sampling_results])
            bbox_targets[pos_rows] = head.bbox_coder.encode(pos_b, torch.cat([r.pos_gt_bboxes for r in sampling_results]))
            bbox_weights.index_fill_(0, pos_rows, 1.0)
        return labels, pos_b.new_ones(N), bbox_targets, bbox_weights

    def _refine(self, rois, bbox_pred, sampling_results, img_metas):
        """BBoxHead.refine_bboxes (bbox_head.py:227-304), class-agnostic: one decode for the whole batch (per-row
        image limits), then per image the rows that were ground truth -- they lead each image's block because gt
        candidates come first and sampled indices are ascending -- are dropped with a slice."""
        head = self.bbox_head[0]
        if not head.reg_class_agnostic or not head.bbox_coder.clip_border:
            return None
        from ..core.bbox import delta2bbox
        boxes = delta2bbox(rois[:, 1:], bbox_pred, head.bbox_coder.means, head.bbox_coder.stds, None)
        lim = const_tensor([[m['img_shape'][1], m['img_shape'][0]] * 2 for m in img_metas], boxes.device, boxes.dtype)      # (B,4) w,h,w,h
        boxes = torch.min(boxes.clamp(min=0), lim[rois[:, 0].long()])
        out, start = [], 0
        if all(hasattr(r, 'num_pos_gt') for r in sampling_results):
            n_gt = [r.num_pos_gt for r in sampling_results]
        else:
            n_gt = [int(v) for v in torch.stack([r.pos_is_gt.sum() for r in sampling_results]).tolist()]
        for r, g in zip(sampling_results, n_gt):
            n = r.pos_bboxes.size(0) + r.neg_bboxes.size(0)
            out.append(boxes[start + g:start + n])
            start += n
        return out

    def _assign_and_sample(self, stage, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore):
        """Per-image reference order when a permutation source is installed (parity tests replay the CPU RNG);
        otherwise all images at once with a single device->host copy (core.bbox.batched_assign_and_sample)."""
        from ..core import bbox as _bbox
        a = self.bbox_assigner[stage]
        if _bbox._randperm is _bbox._device_randperm and all(g is None for g in gt_bboxes_ignore) and \
                a.ignore_iof_thr <= 0 and isinstance(a.neg_iou_thr, float) and \
                type(self.bbox_sampler[stage]).__name__ == 'RandomSampler':
            return _bbox.batched_assign_and_sample(a, self.bbox_sampler[stage], proposal_list, gt_bboxes, gt_labels)[0]
        out = []
        for j in range(len(proposal_list)):
            assign_result = self.bbox_assigner[stage].assign(proposal_list[j], gt_bboxes[j], gt_bboxes_ignore[j],
                                                             gt_labels[j])
            out.append(self.bbox_sampler[stage].sample(assign_result, proposal_list[j], gt_bboxes[j], gt_labels[j]))
        return out

    # ------------------------------------------------------------------ train
    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None):
        losses = dict()
        num_imgs = len(img_metas)
        if gt_bboxes_ignore is None:
            gt_bboxes_ignore = [None for _ in range(num_imgs)]
        sampling_results = self._assign_and_sample(0, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore)
        global_feat = None
        if self.with_global:
            mc_pred, global_feat = self.glbctx_head(x)
            losses['loss_global'] = self.glbctx_head.loss(mc_pred, gt_labels)
        # ---------------- stage 1: common head
        lw = self.stage_loss_weights[0]
        taps = M.PyramidTaps(x[:self.bbox_roi_extractor[0].num_inputs])
        res = self._bbox_forward_train(0, x, sampling_results, gt_bboxes, gt_labels, self.train_cfg[0], img_metas,
                                       global_feat, taps)
        for name, value in res['loss_bbox'].items():
            losses[f's0.{name}'] = value * lw if 'loss' in name else value
        with torch.no_grad():
            roi_labels = res['bbox_targets'][0]
            roi_labels = torch.where(roi_labels == self.bbox_head[0].num_classes,
                                     res['cls_score'][:, :-1].argmax(1), roi_labels)
            proposal_list = self._refine(res['rois'], res['bbox_pred'], sampling_results, img_metas)
            if proposal_list is None:
                proposal_list = self.bbox_head[0].refine_bboxes(res['rois'], roi_labels, res['bbox_pred'],
                                                                [r.pos_is_gt for r in sampling_results], img_metas)
        # ---------------- stage 2: graph reasoning
        lw = self.stage_loss_weights[1]
        sampling_results = self._assign_and_sample(1, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore)
        res = self._bbox_forward_train(1, x, sampling_results, gt_bboxes, gt_labels, self.train_cfg[1], img_metas,
                                       global_feat, taps)
        for name, value in res['loss_bbox'].items():
            losses[f's1.{name}'] = value * lw if 'loss' in name else value
        return losses

    # ------------------------------------------------------------------ train, static shapes
    def can_train_static(self, gt_bboxes_ignore=None):
        """The sync-free training path covers the HTD configs: MaxIoUAssigner without ignore regions and
        RandomSampler in both stages, a class-agnostic stage-1 regressor."""
        from ..core import bbox as _bbox
        if not getattr(self, 'static_shapes', True) or _bbox._randperm is not _bbox._device_randperm:
            return False
        if gt_bboxes_ignore is not None and any(g is not None for g in gt_bboxes_ignore):
            return False
        for a, smp in zip(self.bbox_assigner, self.bbox_sampler):
            if a.ignore_iof_thr > 0 or not isinstance(a.neg_iou_thr, float) or type(smp).__name__ != 'RandomSampler':
                return False
        h = self.bbox_head[0]
        return h.reg_class_agnostic and h.bbox_coder.clip_border and not h.reg_decoded_bbox and \
            all(c.pos_weight <= 0 for c in self.train_cfg)

    def _static_targets(self, stage, S):
        """bbox_head.get_targets (bbox_head.py:85-146) on fixed slots: unused slots carry weight 0."""
        head = self.bbox_head[stage]
        from ..core.bbox import roi_targets_device
        return roi_targets_device(S.boxes.view(-1, 4), S.pos_gt_bboxes.view(-1, 4), S.pos_gt_labels.view(-1),
                                  S.is_pos.view(-1), S.valid.view(-1), head.num_classes, head.bbox_coder.means,
                                  head.bbox_coder.stds)

    def forward_train_static(self, x, img_metas, proposals, n_keep, gt_bboxes, gt_labels):
        """forward_train (htd_roi_head.py:240-349) on fixed-size tensors: proposals (B,P,5) zero-padded past
        n_keep (B,) [device].  Numerically the per-image path with the same samples.  The only host read is the
        number of stage-2 positives, fetched asynchronously behind queued device work."""
        from ..core.bbox import delta2bbox_clip_device, static_assign_and_sample
        losses = dict()
        B, P = proposals.shape[:2]
        dev = proposals.device
        pvalid = arange_cached(P, dev)[None, :] < n_keep[:, None]
        S0 = static_assign_and_sample(self.bbox_assigner[0], self.bbox_sampler[0], proposals[..., :4], pvalid,
                                      gt_bboxes, gt_labels)
        global_feat = None
        if self.with_global:
            mc_pred, global_feat = self.glbctx_head(x)
            losses['loss_global'] = self.glbctx_head.loss(mc_pred, gt_labels)
        # ---------------- stage 1: common head
        rois = S0.rois
        taps = M.PyramidTaps(x[:self.bbox_roi_extractor[0].num_inputs])
        res = self._bbox_forward(0, x, rois, global_feat, taps=taps)
        t0 = self._static_targets(0, S0)
        loss0 = self.bbox_head[0].loss(res['cls_score'], res['bbox_pred'], rois, *t0, num_samples=S0.valid.sum())
        lw = self.stage_loss_weights[0]
        for name, value in loss0.items():
            losses[f's0.{name}'] = value * lw if 'loss' in name else value
        with torch.no_grad():            # refine_bboxes (bbox_head.py:227-304): decode, clip, drop the gt-born rows
            head = self.bbox_head[0]
            lim = const_tensor([[m['img_shape'][1], m['img_shape'][0]] for m in img_metas], dev, torch.float32)
            n = S0.valid.size(1)
            keep = S0.valid & ~S0.pos_is_gt
            boxes = delta2bbox_clip_device(S0.boxes.view(-1, 4), res['bbox_pred'], head.bbox_coder.means,
                                           head.bbox_coder.stds, lim, keep.view(-1), n).view(B, n, 4)
        # ---------------- stage 2: graph reasoning
        S1 = static_assign_and_sample(self.bbox_assigner[1], self.bbox_sampler[1], boxes, keep, gt_bboxes, gt_labels)
        rois = S1.rois
        n = S1.valid.size(1)
        extractor, enhanced_extractor = self.bbox_roi_extractor[0], self.bbox_roi_extractor[1]
        feats = taps
        bbox_feats = extractor(feats, rois)
        # The regression branch runs on the positives only.  Their count is the one number of the step the host
        # needs: it is copied to pinned memory asynchronously now and read AFTER the classification branch has been
        # queued, so the device still has that work to do while the host waits for the copy (no idle gap), and the
        # host never waits for more than the sampling kernels.
        npos_host = torch.empty(B, dtype=S1.npos.dtype).pin_memory()
        npos_host.copy_(S1.npos, non_blocking=True)
        npos_ready = torch.cuda.Event()
        npos_ready.record()
        head = self.bbox_head[1]
        gf = global_feat if self.with_global else None
        # the positives' rows of bbox_feats are read after the classification branch is queued: through the alias that branch's
        # first node leaves in the stash, so that their gradient joins its sum in place (mmcv_ops.PlainAndFusedFunction)
        stash = M.RowStash() if (gf is not None and bbox_feats.is_cuda and torch.is_grad_enabled()) else None
        cls_score = head.forward_cls(bbox_feats, feats, rois, self.bbox_head[0].fc_cls, gf, rois_per_img=(n, ) * B,
                                     roi_valid=S1.valid.view(-1), row_stash=stash)
        t1 = self._static_targets(1, S1)          # needs no count: queued before the host stops for it
        full = cls_score.new_zeros(cls_score.size(0), 4)
        nvalid1 = S1.valid.sum()
        npos_ready.synchronize()
        npos = [int(v) for v in npos_host.tolist()]
        if sum(npos) > 0:
            # The count changes from step to step, and with it the size of every tensor of the regression branch: the caching
            # allocator then keeps a block per size it has seen (+277 MB of reserve per 500 steps with the optimizer
            # on).  The row list is therefore padded to a multiple of POS_BUCKET with slots that are NOT positives (the first
            # image's next slots: negatives or unused ones); their predictions land in rows whose regression weight is zero
            # (bbox_head.py:165-183), so losses and gradients are unchanged and the sizes repeat.
            take = list(npos)
            pad = (-sum(npos)) % POS_BUCKET
            for b in range(B):
                extra = min(pad, n - take[b])
                take[b] += extra
                pad -= extra
            pos_rows = torch.cat([torch.arange(b * n, b * n + k, device=dev) for b, k in enumerate(take)])
            pos_rois = torch.index_select(rois, 0, pos_rows)
            enhanced = enhanced_extractor(feats, pos_rois)
            pos_feats = M.select_rows_via(stash, pos_rows) if (stash is not None and stash.alias is not None) else \
                torch.index_select(bbox_feats, 0, pos_rows)
            bbox_pred = head.forward_reg(pos_feats, enhanced, pos_rois, gf)
            full = full.index_copy(0, pos_rows, bbox_pred)
        # no positive in the whole batch: the regression branch gets no gradient this step (zeros in the flat buffer)
        loss1 = self.bbox_head[1].loss(cls_score, full, rois, *t1, num_samples=nvalid1)
        lw = self.stage_loss_weights[1]
        for name, value in loss1.items():
            losses[f's1.{name}'] = value * lw if 'loss' in name else value
        self._last_static = (S0, S1)          # exposed for tests
        return losses

    # ------------------------------------------------------------------ test
    batched_test = True      # post-process the whole batch in one pass (False: the per-image loop of the reference)

    def _batched_test_ok(self, rois, img_metas, rescale):
        if not (self.batched_test and rois.is_cuda and len(img_metas) > 1):
            return False
        if not all(getattr(h, 'with_reg', False) for h in self.bbox_head):
            return False                                            # get_bboxes without deltas clips by scalar img_shape
        if dict(self.test_cfg.nms).get('type', 'nms') != 'nms':
            return False                                            # soft-NMS decays sequentially per class: per image
        kinds = {isinstance(m['scale_factor'], float) for m in img_metas}
        return not rescale or len(kinds) == 1

    def simple_test_bboxes(self, x, proposal_list, img_metas, rescale=False):
        """-> (det_bboxes list, det_labels list) on the device."""
        num_imgs = len(proposal_list)
        rois = bbox2roi(proposal_list)
        global_feat = self.glbctx_head(x)[1] if self.with_global else None
        n_per = tuple(len(p) for p in proposal_list)
        res = self._bbox_forward(0, x, rois, global_feat)
        cls0, reg0 = res['cls_score'], res['bbox_pred']
        # stage-1 refinement with the arg-max foreground class (:346-352); class-agnostic => label unused
        label = cls0[:, :-1].argmax(dim=1)
        batched = self._batched_test_ok(rois, img_metas, rescale)
        if batched:
            # every row carries its image's clip limits / scale: the same arithmetic as the per-image calls, one pass
            img_of = rois[:, 0].long()
            hw = torch.tensor([[float(m['img_shape'][0]), float(m['img_shape'][1])] for m in img_metas],
                              dtype=rois.dtype).to(rois.device, non_blocking=True)[img_of]
            rois = self.bbox_head[0].regress_by_class(rois, label, reg0, dict(img_shape=hw))
        else:
            rois = torch.cat([self.bbox_head[0].regress_by_class(r, l, p, m) for r, l, p, m in
                              zip(rois.split(n_per), label.split(n_per), reg0.split(n_per), img_metas)])
        res = self._bbox_forward(1, x, rois, global_feat)
        cls_score = (cls0 + res['cls_score']) / 2.0              # logits averaged over the stages (:363-366)
        if batched:
            return self._get_bboxes_images(rois, cls_score, res['bbox_pred'], img_of, hw, img_metas, rescale)
        det_bboxes, det_labels = [], []
        for i, (r, c, p) in enumerate(zip(rois.split(n_per), cls_score.split(n_per), res['bbox_pred'].split(n_per))):
            b, l = self.bbox_head[-1].get_bboxes(r, c, p, img_metas[i]['img_shape'], img_metas[i]['scale_factor'],
                                                 rescale=rescale, cfg=self.test_cfg)
            det_bboxes.append(b)
            det_labels.append(l)
        return det_bboxes, det_labels

    def _get_bboxes_images(self, rois, cls_score, bbox_pred, img_of, hw, img_metas, rescale):
        """BBoxHead.get_bboxes (bbox_heads/bbox_head.py:309-341) of every image at once; results equal the per-image calls
        bit for bit (tests/test_gpu_detector.py::test_batched_test_postprocessing_equals_the_per_image_loop)."""
        from ..core.post_processing import multiclass_nms_images
        head = self.bbox_head[-1]
        bboxes, scores = head.get_bboxes(rois, cls_score, bbox_pred, hw, None, rescale=False, cfg=None)
        if rescale and bboxes.size(0) > 0:
            if isinstance(img_metas[0]['scale_factor'], float):
                # tensor / python scalar multiplies by the fp32 reciprocal on the device; same here, per row
                inv = (1.0 / torch.tensor([m['scale_factor'] for m in img_metas], dtype=torch.float32))
                bboxes = bboxes * inv.to(bboxes.device, non_blocking=True)[img_of][:, None]
            else:
                sf = torch.tensor([[float(v) for v in m['scale_factor']] for m in img_metas], dtype=torch.float32)
                sf = sf.to(bboxes.device, non_blocking=True)[img_of]
                bboxes = (bboxes.view(bboxes.size(0), -1, 4) / sf[:, None, :]).view(bboxes.size(0), -1)
        return multiclass_nms_images(bboxes, scores, img_of, len(img_metas), self.test_cfg.score_thr, self.test_cfg.nms,
                                     self.test_cfg.max_per_img)

    def simple_test(self, x, proposal_list, img_metas, rescale=False):
        from ..core.bbox import bbox2result_many
        det_bboxes, det_labels = self.simple_test_bboxes(x, proposal_list, img_metas, rescale)
        if self.batched_test:
            return bbox2result_many(det_bboxes, det_labels, self.bbox_head[-1].num_classes)
        return [bbox2result(b, l, self.bbox_head[-1].num_classes) for b, l in zip(det_bboxes, det_labels)]

    def aug_test(self, features, proposal_list, img_metas, rescale=False):
        """roi_heads/htd_roi_head.py:388-433: the merged proposals of one image go through both stages on every
        augmentation's pyramid; boxes are mapped back, averaged with the scores, and take ONE multi-class NMS.
        Like the reference, the result is in the original image scale whatever `rescale` says."""
        from ..core.bbox import bbox_mapping
        from ..core.post_processing import merge_aug_bboxes, multiclass_nms
        aug_bboxes, aug_scores = [], []
        for x, img_meta in zip(features, img_metas):
            m = img_meta[0]                                         # one image per batch in aug test
            proposals = bbox_mapping(proposal_list[0][:, :4], m['img_shape'], m['scale_factor'], m['flip'],
                                     m['flip_direction'])
            global_feat = self.glbctx_head(x)[1] if self.with_global else None
            rois = bbox2roi([proposals])
            ms_scores = []
            for i in range(self.num_stages):
                res = self._bbox_forward(i, x, rois, global_feat)
                ms_scores.append(res['cls_score'])
                if i < self.num_stages - 1:
                    label = res['cls_score'][:, :-1].argmax(dim=1)
                    rois = self.bbox_head[i].regress_by_class(rois, label, res['bbox_pred'], m)
            cls_score = sum(ms_scores) / float(len(ms_scores))
            bboxes, scores = self.bbox_head[-1].get_bboxes(rois, cls_score, res['bbox_pred'], m['img_shape'],
                                                           m['scale_factor'], rescale=False, cfg=None)
            aug_bboxes.append(bboxes)
            aug_scores.append(scores)
        merged_bboxes, merged_scores = merge_aug_bboxes(aug_bboxes, aug_scores, img_metas, self.test_cfg)
        det_bboxes, det_labels = multiclass_nms(merged_bboxes, merged_scores, self.test_cfg.score_thr, self.test_cfg.nms,
                                                self.test_cfg.max_per_img)
        return [bbox2result(det_bboxes, det_labels, self.bbox_head[-1].num_classes)]
