"""RPN head of HTD: forward, targets + loss, proposal generation.

Reference: dense_heads/rpn_head.py:12-168 (RPNHead), anchor_head.py:14-682 (AnchorHead: targets/loss/
get_bboxes), base_dense_head.py:22-59 (forward_train), rpn_test_mixin.py:24-37 (simple_test_rpn).
Same registry name, constructor kwargs, state_dict keys (rpn_conv / rpn_cls / rpn_reg) and return
structures.  What differs is execution: proposal generation handles ALL images and levels of the batch
together -- one stable sort per level over the (B, A_l) score matrix, one decode, ONE batched NMS
launch whose segments are the (image, level) pairs -- instead of a Python loop over images with a
sort + NMS each (rpn_head.py:78-168).  The kept set, its order and its values are the same.
"""
import torch
import torch.nn as nn

from ..core import (anchor_inside_flags, images_to_levels, multi_apply, unmap)
from ..core.bbox import delta2bbox, delta2bbox_clip_device
from ..core.misc import arange_cached, const_tensor
from .. import mmcv_ops as M
from ..mmcv_ops import nms_sorted_mask
from ..registry import (HEADS, build_anchor_generator, build_assigner, build_bbox_coder, build_loss, build_sampler)
from .bricks import Conv2d, normal_init


class _RPNLossFunction(torch.autograd.Function):
    """(sum of weighted BCE, sum of SmoothL1 on positives) over every anchor of the batch: one launch forward
    (htd_rpn_loss, which also leaves the derivatives), two scalings backward."""

    @staticmethod
    def forward(ctx, cls, reg, anchors, gts, assigned, pos, neg, means, stds, beta, pos_weight):
        from .. import capi
        from ..core.bbox import _f4
        B, A = assigned.shape
        K = gts.size(1)
        cls, reg = cls.float().contiguous(), reg.float().contiguous()
        anchors, gts = anchors.float().contiguous(), gts.float().contiguous()
        assigned = assigned.contiguous()
        pos8, neg8 = pos.to(torch.uint8).contiguous(), neg.to(torch.uint8).contiguous()
        partial = torch.empty(capi.lib().htd_rpn_loss_partial_rows(), 2, device=cls.device, dtype=torch.float32)
        gcls, greg = torch.empty_like(cls), torch.empty_like(reg)
        capi.call('htd_rpn_loss', capi.ptr(cls), capi.ptr(reg), capi.ptr(anchors), capi.ptr(gts), capi.ptr(assigned),
                  capi.ptr(pos8), capi.ptr(neg8), B, A, K, _f4(means), _f4(stds), beta, pos_weight, capi.ptr(partial),
                  capi.ptr(gcls), capi.ptr(greg), capi.current_stream_ptr())
        ctx.save_for_backward(gcls, greg)
        sums = partial.sum(0)
        return sums[0], sums[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_cls, g_box):
        gcls, greg = ctx.saved_tensors
        return (gcls * g_cls, greg * g_box) + (None, ) * 9


class _SplitHeads(torch.autograd.Function):
    """(B, nc + nr + pad, H, W) -> channel slices [:nc], [nc:nc+nr] (views).  Backward assembles the gradient of the merged
    map in ONE concatenation instead of two zero-filled slice scatters and their sum."""

    @staticmethod
    def forward(ctx, y, nc, nr):
        ctx.dims = (nc, nr, y.size(1))
        ctx.set_materialize_grads(False)
        return y[:, :nc], y[:, nc:nc + nr]

    @staticmethod
    def backward(ctx, gc, gr):
        nc, nr, C = ctx.dims
        ref = gc if gc is not None else gr
        if ref is None:
            return None, None, None
        B, _, H, W = ref.shape
        parts = [gc if gc is not None else ref.new_zeros(B, nc, H, W), gr if gr is not None else ref.new_zeros(B, nr, H, W)]
        if C > nc + nr:
            parts.append(ref.new_zeros(B, C - nc - nr, H, W))
        # (B, H, W, C) is the memory order of a channels_last map: concatenate there, hand back the NCHW view of it
        g = torch.cat([p.permute(0, 2, 3, 1) for p in parts], 3).permute(0, 3, 1, 2)
        return g, None, None


RPN_FLAT_HEADS = __import__('os').environ.get('HTD_RPN_FLAT', '1') != '0'       # 0: per-level split / reshape / cat (A/B runs)


class LevelList(list):
    """The per-level (B, A*C, h, w) maps the reference's head API passes around (anchor_head.py:123-140), as VIEWS of one flat
    per-anchor tensor `.flat` -- (B, A_total) objectness or (B, A_total, 4) deltas, level-major, the order
    permute(0, 2, 3, 1).reshape + cat gives -- which the batched loss and the proposal stage read directly."""
    flat = None


class _RPNGatherHeads(torch.autograd.Function):
    """merged head outputs y_l (B, C, h_l, w_l) channels_last of all levels -> (cls (B, A), reg (B, A, 4)) in ONE launch
    (htd_rpn_heads_gather); backward: the gradients of every y_l, written in full by one launch (htd_rpn_heads_scatter) --
    instead of a channel split, two reshape copies per level, two concatenations and their mirror images in backward."""

    @staticmethod
    def forward(ctx, na, *ys):
        import ctypes
        from .. import capi
        ys = [y.contiguous(memory_format=torch.channels_last) for y in ys]
        B, C = ys[0].size(0), ys[0].size(1)
        L = len(ys)
        pix = [y.size(2) * y.size(3) for y in ys]
        A = na * sum(pix)
        cls = torch.empty(B, A, device=ys[0].device, dtype=torch.float32)
        reg = torch.empty(B, A, 4, device=ys[0].device, dtype=torch.float32)
        capi.call('htd_rpn_heads_gather', (ctypes.c_void_p * L)(*[y.data_ptr() for y in ys]), (ctypes.c_int64 * L)(*pix), L, B, C,
                  int(na), capi.ptr(cls), capi.ptr(reg), capi.current_stream_ptr())
        ctx.meta = (int(na), [tuple(y.shape) for y in ys])
        ctx.set_materialize_grads(False)
        return cls, reg

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gcls, greg):
        import ctypes
        from .. import capi
        na, shapes = ctx.meta
        if gcls is None and greg is None:
            return (None, ) * (1 + len(shapes))
        B, C = shapes[0][0], shapes[0][1]
        L = len(shapes)
        pix = [s_[2] * s_[3] for s_ in shapes]
        A = na * sum(pix)
        ref = gcls if gcls is not None else greg
        gcls = gcls.contiguous() if gcls is not None else ref.new_zeros(B, A)
        greg = greg.contiguous() if greg is not None else ref.new_zeros(B, A, 4)
        gys = [torch.empty(s_, device=ref.device, dtype=torch.float32, memory_format=torch.channels_last) for s_ in shapes]
        capi.call('htd_rpn_heads_scatter', capi.ptr(gcls), capi.ptr(greg), (ctypes.c_void_p * L)(*[g.data_ptr() for g in gys]),
                  (ctypes.c_int64 * L)(*pix), L, B, C, na, capi.current_stream_ptr())
        return (None, *gys)


@HEADS.register_module()
class RPNHead(nn.Module):
    def __init__(self, in_channels, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator', scales=[8, 16, 32], ratios=[0.5, 1.0, 2.0],
                                       strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=(.0, .0, .0, .0),
                                 target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0), train_cfg=None, test_cfg=None):
        super().__init__()
        self.in_channels, self.num_classes, self.feat_channels = in_channels, 1, feat_channels
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        assert self.use_sigmoid_cls and not reg_decoded_bbox, 'HTD configs use a sigmoid RPN with delta targets'
        self.sampling = True
        self.cls_out_channels = self.num_classes
        self.reg_decoded_bbox = reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            self.sampler = build_sampler(self.train_cfg.sampler, context=self)
        self.anchor_generator = build_anchor_generator(anchor_generator)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self.rpn_conv = Conv2d(self.in_channels, self.feat_channels, 3, padding=1)
        self.rpn_cls = Conv2d(self.feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.rpn_reg = Conv2d(self.feat_channels, self.num_anchors * 4, 1)

    def init_weights(self):
        normal_init(self.rpn_conv, std=0.01)
        normal_init(self.rpn_cls, std=0.01)
        normal_init(self.rpn_reg, std=0.01)

    # ------------------------------------------------------------------ forward
    def forward_single(self, x):
        x = self.rpn_conv(x, relu=True)
        if x.dtype != torch.float32:             # bf16 pyramid: the 3- and 12-channel heads and all box math stay fp32
            x = x.float()
        return self.rpn_cls(x), self.rpn_reg(x)

    def forward(self, feats):
        if not feats[0].is_cuda:
            return multi_apply(self.forward_single, feats)
        # the two 1x1 heads read the same hidden map: as ONE 16-channel convolution (3 + 12 + a zero row) the hidden map gets
        # one data gradient instead of two that autograd has to add, and every level is one launch less in each direction
        nc, nr = self.rpn_cls.out_channels, self.rpn_reg.out_channels
        pad = -(nc + nr) % 8
        w = torch.cat([self.rpn_cls.weight, self.rpn_reg.weight] +
                      ([self.rpn_cls.weight.new_zeros(pad, *self.rpn_cls.weight.shape[1:])] if pad else []))
        b = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias] + ([self.rpn_cls.bias.new_zeros(pad)] if pad else []))
        cls, reg, merged = [], [], []
        taps = isinstance(feats, M.PyramidTaps)      # the pyramid as a chain of consumers: see Conv2dFunction(chain=True)
        from .. import dense
        if torch.is_grad_enabled() and w.requires_grad:
            # five levels read these two tensors: their gradients collect in one buffer each (dense.temp_grad_sink)
            dense.temp_grad_sink(w)
            dense.temp_grad_sink(b)
        for i in range(len(feats)):
            x = feats[i]
            if x.dtype == torch.float32 and x.size(1) % 8 == 0:
                # 3x3 + ReLU + merged 1x1 heads as one autograd node (the hidden map's ReLU mask rides the head's dgrad)
                y = dense.conv_relu_head(x, self.rpn_conv.weight, self.rpn_conv.bias, w, b, self.rpn_conv.padding[0], taps)
                if taps:
                    y, feats.levels[i] = y
            else:
                h = self.rpn_conv(x, relu=True)
                if h.dtype != torch.float32:     # bf16 pyramid: the heads and all box math stay fp32
                    h = h.float()
                y = self.rpn_cls(h, weight=w, bias=b)
            merged.append(y)
        if RPN_FLAT_HEADS and self.cls_out_channels == 1 and nr == 4 * nc and all(y.dtype == torch.float32 for y in merged):
            # all levels -> the flat per-anchor tensors in one launch; the per-level maps of the head API are views of them
            cls_all, reg_all = _RPNGatherHeads.apply(nc, *merged)
            cls, reg, off = LevelList(), LevelList(), 0
            B = cls_all.size(0)
            for y in merged:
                hh, ww = y.size(2), y.size(3)
                n = hh * ww * nc
                cls.append(cls_all[:, off:off + n].view(B, hh, ww, nc).permute(0, 3, 1, 2))
                reg.append(reg_all[:, off:off + n].reshape(B, hh, ww, nr).permute(0, 3, 1, 2))
                off += n
            cls.flat, reg.flat = cls_all, reg_all
            return cls, reg
        for y in merged:
            c, r = _SplitHeads.apply(y, nc, nr)
            cls.append(c)
            reg.append(r)
        return cls, reg

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      **kwargs):
        outs = self(x)
        losses = self.loss(*outs, gt_bboxes, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is None:
            return losses
        return losses, self.get_bboxes(*outs, img_metas, cfg=proposal_cfg, padded=kwargs.get('padded', False))

    def simple_test_rpn(self, x, img_metas):
        return self.get_bboxes(*self(x), img_metas)

    def aug_test_rpn(self, feats, img_metas):
        """dense_heads/rpn_test_mixin.py:39-59: proposals of every augmentation, merged per image at original scale."""
        from ..core.post_processing import merge_aug_proposals
        samples_per_gpu = len(img_metas[0])
        aug_proposals = [[] for _ in range(samples_per_gpu)]
        for x, img_meta in zip(feats, img_metas):
            for i, proposals in enumerate(self.simple_test_rpn(x, img_meta)):
                aug_proposals[i].append(proposals)
        aug_img_metas = [[img_metas[j][i] for j in range(len(img_metas))] for i in range(samples_per_gpu)]
        return [merge_aug_proposals(p, m, self.test_cfg) for p, m in zip(aug_proposals, aug_img_metas)]

    # ------------------------------------------------------------------ targets + loss
    def get_anchors(self, featmap_sizes, img_metas, device='cuda'):
        mlvl = self.anchor_generator.grid_anchors(featmap_sizes, device)
        anchor_list = [mlvl for _ in img_metas]
        valid_flag_list = [self.anchor_generator.valid_flags(featmap_sizes, m['pad_shape'], device)
                           for m in img_metas]
        return anchor_list, valid_flag_list

    def _get_targets_single(self, flat_anchors, valid_flags, gt_bboxes, gt_bboxes_ignore, img_meta):
        """anchor_head.py:172-269 with gt_labels=None (RPN): fg label 0, bg label num_classes (=1)."""
        inside = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2],
                                     self.train_cfg.allowed_border)
        anchors = flat_anchors[inside, :]
        assign_result = self.assigner.assign(anchors, gt_bboxes, gt_bboxes_ignore, None)
        sr = self.sampler.sample(assign_result, anchors, gt_bboxes)
        n = anchors.shape[0]
        bbox_targets = torch.zeros_like(anchors)
        bbox_weights = torch.zeros_like(anchors)
        labels = anchors.new_full((n, ), self.num_classes, dtype=torch.long)
        label_weights = anchors.new_zeros(n, dtype=torch.float)
        if len(sr.pos_inds) > 0:
            bbox_targets[sr.pos_inds, :] = self.bbox_coder.encode(sr.pos_bboxes, sr.pos_gt_bboxes)
            bbox_weights[sr.pos_inds, :] = 1.0
            labels[sr.pos_inds] = 0
            label_weights[sr.pos_inds] = 1.0 if self.train_cfg.pos_weight <= 0 else self.train_cfg.pos_weight
        if len(sr.neg_inds) > 0:
            label_weights[sr.neg_inds] = 1.0
        total = flat_anchors.size(0)
        return (unmap(labels, total, inside, fill=self.num_classes), unmap(label_weights, total, inside),
                unmap(bbox_targets, total, inside), unmap(bbox_weights, total, inside), sr.pos_inds, sr.neg_inds)

    def get_targets(self, anchor_list, valid_flag_list, gt_bboxes_list, img_metas, gt_bboxes_ignore_list=None):
        num_imgs = len(img_metas)
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        flat_anchors = torch.cat(anchor_list[0])
        if gt_bboxes_ignore_list is None:
            gt_bboxes_ignore_list = [None] * num_imgs
        res = [self._get_targets_single(flat_anchors, torch.cat(valid_flag_list[i]), gt_bboxes_list[i],
                                        gt_bboxes_ignore_list[i], img_metas[i]) for i in range(num_imgs)]
        num_total_pos = sum(max(r[4].numel(), 1) for r in res)
        num_total_neg = sum(max(r[5].numel(), 1) for r in res)
        lv = [images_to_levels([r[k] for r in res], num_level_anchors) for k in range(4)]
        return lv[0], lv[1], lv[2], lv[3], num_total_pos, num_total_neg

    def loss_single(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_total_samples):
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = self.loss_cls(cls_score, labels.reshape(-1), label_weights.reshape(-1),
                                 avg_factor=num_total_samples)
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, 4)
        loss_bbox = self.loss_bbox(bbox_pred, bbox_targets.reshape(-1, 4), bbox_weights.reshape(-1, 4),
                                   avg_factor=num_total_samples)
        return loss_cls, loss_bbox

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None):
        """Exact reference order of operations (per image, per level) when a permutation source is installed
        (`core.set_randperm`, used by parity tests to replay the CPU generator); otherwise the batched,
        host-sync-free formulation below, which computes the same quantities for the whole batch at once."""
        from ..core import bbox as _bbox
        if _bbox._randperm is _bbox._device_randperm and gt_bboxes_ignore is None and \
                self.assigner.ignore_iof_thr <= 0 and isinstance(self.assigner.neg_iou_thr, float):
            return self.loss_batched(cls_scores, bbox_preds, gt_bboxes, img_metas)
        return self.loss_per_image(cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore)

    def _anchors_inside(self, featmap_sizes, img_metas, dev):
        """(A,4) level-concatenated anchors and the (B,A) mask of anchors that are valid and inside their image
        (anchor_head.py:200-207, core/anchor/utils.py:20-46): constants of (feature sizes, image shapes), cached."""
        border = self.train_cfg.allowed_border
        key = (tuple(tuple(int(v) for v in f) for f in featmap_sizes),
               tuple((tuple(m['img_shape'][:2]), tuple(m['pad_shape'][:2])) for m in img_metas), str(dev), border)
        cache = self.__dict__.setdefault('_inside_cache', {})
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, device=dev)
            flat_anchors = torch.cat(anchor_list[0])
            valid = torch.stack([torch.cat(v) for v in valid_flag_list])                   # (B,A)
            lim = const_tensor([[m['img_shape'][1], m['img_shape'][0]] for m in img_metas], dev, flat_anchors.dtype)
            if border >= 0:
                inside = valid & (flat_anchors[None, :, 0] >= -border) & (flat_anchors[None, :, 1] >= -border) & \
                    (flat_anchors[None, :, 2] < lim[:, 0:1] + border) & (flat_anchors[None, :, 3] < lim[:, 1:2] + border)
            else:
                inside = valid
            cache[key] = (flat_anchors, inside)
        return cache[key]

    # -------------------------------------------------------------- batched targets + loss (production path)
    def loss_batched(self, cls_scores, bbox_preds, gt_bboxes, img_metas, keys=None):
        dev = cls_scores[0].device
        B = cls_scores[0].size(0)
        featmap_sizes = [f.size()[-2:] for f in cls_scores]
        flat_anchors, inside = self._anchors_inside(featmap_sizes, img_metas, dev)
        A = flat_anchors.size(0)
        from ..core.bbox import pad_gt_batch
        gts, gt_valid = pad_gt_batch(gt_bboxes)
        from ..core.bbox import (_sample_on_device, batched_max_iou_assign, batched_random_sample, random_sample_device,
                                 sample_keys)
        assigned, _ = batched_max_iou_assign(self.assigner, flat_anchors, inside, gts, gt_valid)
        sc = self.train_cfg.sampler
        if keys is None:
            keys = sample_keys(flat_anchors[None].expand(B, A, 4))
        if _sample_on_device(assigned, keys, sc.num):       # masks and drawn counts from one call (htd_random_sample)
            pos, neg, counts, _ = random_sample_device(assigned, keys, sc.num, sc.pos_fraction, sc.get('neg_pos_ub', -1))
            n_pos, n_neg = counts[:, 0], counts[:, 1]
        else:
            pos, neg = batched_random_sample(assigned, sc.num, sc.pos_fraction, sc.get('neg_pos_ub', -1), keys)
            n_pos, n_neg = pos.sum(1), neg.sum(1)
        num_total = (n_pos.clamp(min=1) + n_neg.clamp(min=1)).sum().to(torch.float32)       # anchor_head.py:354-355
        if getattr(cls_scores, 'flat', None) is not None and getattr(bbox_preds, 'flat', None) is not None:
            cls, reg = cls_scores.flat.view(B, -1, self.cls_out_channels), bbox_preds.flat      # (RPNHead.forward made them)
        else:
            cls = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels) for c in cls_scores], 1)
            reg = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4) for r in bbox_preds], 1)
        self._last_rpn_sample = (assigned, pos, neg, inside)  # exposed for tests
        if self._fused_loss_ok():
            s_cls, s_box = _RPNLossFunction.apply(cls.reshape(-1), reg.reshape(-1, 4), flat_anchors, gts, assigned,
                                                  pos, neg, tuple(self.bbox_coder.means), tuple(self.bbox_coder.stds),
                                                  float(self.loss_bbox.beta), float(self.train_cfg.pos_weight))
            return dict(loss_rpn_cls=[self.loss_cls.loss_weight * s_cls / num_total],
                        loss_rpn_bbox=[self.loss_bbox.loss_weight * s_box / num_total])
        # targets (anchor_head.py:172-269): labels 0 = foreground, num_classes = background; weights 1 on samples
        gt_of = torch.gather(gts, 1, (assigned - 1).clamp(min=0)[..., None].expand(B, A, 4))
        safe_gt = torch.where(pos[..., None], gt_of, flat_anchors[None].expand(B, A, 4))   # avoid log(0) on unused rows
        bbox_targets = self.bbox_coder.encode(flat_anchors[None].expand(B, A, 4).reshape(-1, 4),
                                              safe_gt.reshape(-1, 4)).view(B, A, 4)
        posf = pos.to(bbox_targets.dtype)
        label_weights = (pos | neg).to(bbox_targets.dtype)
        if self.train_cfg.pos_weight > 0:
            label_weights = torch.where(pos, label_weights.new_full((1, ), self.train_cfg.pos_weight), label_weights)
        labels = torch.where(pos, torch.zeros_like(assigned), torch.full_like(assigned, self.num_classes))
        # one pass over all levels: sum_l loss_l == loss on the level-concatenated tensors (same avg_factor)
        loss_cls = self.loss_cls(cls.reshape(-1, self.cls_out_channels), labels.reshape(-1), label_weights.reshape(-1),
                                 avg_factor=num_total)
        loss_bbox = self.loss_bbox(reg.reshape(-1, 4), (bbox_targets * posf[..., None]).reshape(-1, 4),
                                   posf[..., None].expand(B, A, 4).reshape(-1, 4), avg_factor=num_total)
        return dict(loss_rpn_cls=[loss_cls], loss_rpn_bbox=[loss_bbox])

    def _fused_loss_ok(self):
        """htd_rpn_loss covers the RPN of every HTD config: one sigmoid channel, BCE without class weights,
        SmoothL1 on encoded deltas."""
        lc, lb = self.loss_cls, self.loss_bbox
        return getattr(self, 'fused_loss', True) and self.cls_out_channels == 1 and \
            type(lc).__name__ == 'CrossEntropyLoss' and lc.use_sigmoid and lc.class_weight is None and \
            lc.reduction == 'mean' and type(lb).__name__ == 'SmoothL1Loss' and lb.reduction == 'mean' and \
            not getattr(self, 'reg_decoded_bbox', False)

    # -------------------------------------------------------------- reference-order path
    def loss_per_image(self, cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None):
        featmap_sizes = [f.size()[-2:] for f in cls_scores]
        assert len(featmap_sizes) == self.anchor_generator.num_levels
        anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, device=cls_scores[0].device)
        (labels, label_weights, bbox_targets, bbox_weights, num_pos, num_neg) = self.get_targets(
            anchor_list, valid_flag_list, gt_bboxes, img_metas, gt_bboxes_ignore_list=gt_bboxes_ignore)
        losses_cls, losses_bbox = multi_apply(self.loss_single, cls_scores, bbox_preds, labels, label_weights,
                                              bbox_targets, bbox_weights, num_total_samples=num_pos + num_neg)
        return dict(loss_rpn_cls=losses_cls, loss_rpn_bbox=losses_bbox)

    # ------------------------------------------------------------------ proposals
    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False, with_nms=True, padded=False):
        """-> list (per image) of (k_i, 5) [x1,y1,x2,y2,score], k_i <= nms_post, descending score.
        padded=True: -> (dets (B, nms_post, 5) zero rows past k_i, k (B,) on the device) without reading k back."""
        cfg = self.test_cfg if cfg is None else cfg
        assert len(cls_scores) == len(bbox_preds)
        B = cls_scores[0].size(0)
        dev = cls_scores[0].device
        featmap_sizes = [c.shape[-2:] for c in cls_scores]
        mlvl_anchors = None                                 # (made on demand: the flat path keeps its concatenation cached)
        scores_l, deltas_l, anchors_l, seg_sizes = [], [], [], []
        record = getattr(self, 'record_trail', False)       # tests: which candidates survive, in which order
        flat_ids, level_off = [], 0
        L = len(cls_scores)
        Ns = [int(c.shape[1] * c.shape[2] * c.shape[3]) for c in cls_scores]
        ks = [n if cfg.nms_pre <= 0 else min(cfg.nms_pre, n) for n in Ns]
        fused = dev.type == 'cuda' and max(ks) <= M.TOPK_KMAX and cls_scores[0].dtype == torch.float32
        if fused:
            # every (image, level) ranking of the call in one segmented top-k (htd_segmented_topk): the sorted first nms_pre of
            # `scores.sort(descending=True)` (rpn_head.py:122-133), equal scores by ascending anchor index
            total = sum(Ns)
            if getattr(cls_scores, 'flat', None) is not None:
                sig = cls_scores.flat.detach().sigmoid()
            else:
                sig = torch.cat([c.detach().permute(0, 2, 3, 1).reshape(B, -1) for c in cls_scores], 1).sigmoid_()
            offs = [sum(Ns[:l]) for l in range(L)]
            top_idx, top_val = M.segmented_topk(sig, [(b * total + offs[l], Ns[l], ks[l]) for b in range(B) for l in range(L)])
            top_idx, top_val = top_idx.view(B, sum(ks)), top_val.view(B, sum(ks))
        flat_ok = fused and getattr(cls_scores, 'flat', None) is not None and getattr(bbox_preds, 'flat', None) is not None and \
            not record
        if flat_ok:
            # the levels' candidates through ONE gather each: rank inside (image, level) + the level's first anchor = index into
            # the flat per-anchor tensors and the level-concatenated anchors (constants of the map sizes: cached)
            ck = (tuple(Ns), tuple(ks), str(dev))
            cache = self.__dict__.setdefault('_prop_cache', {})
            if ck not in cache:
                if len(cache) > 32:
                    cache.clear()
                mlvl_anchors = self.anchor_generator.grid_anchors(featmap_sizes, device=dev)
                col_off = torch.cat([torch.full((k, ), offs[l], dtype=torch.int64) for l, k in enumerate(ks)]).to(dev)
                lvl_ids = torch.cat([torch.full((k, ), float(l)) for l, k in enumerate(ks)]).to(dev)
                cache[ck] = (col_off, lvl_ids, torch.cat(mlvl_anchors))
            col_off, lvl_ids, all_anchors = cache[ck]
            gidx = top_idx + col_off[None]
            scores_l, seg_sizes = [top_val], list(ks)
            deltas_l = [torch.gather(bbox_preds.flat.detach(), 1, gidx[..., None].expand(B, gidx.size(1), 4))]
            anchors_l = [all_anchors[gidx]]
        koff = 0
        if not flat_ok:
            mlvl_anchors = self.anchor_generator.grid_anchors(featmap_sizes, device=dev)
        for lvl in range(L if not flat_ok else 0):
            d = bbox_preds[lvl].detach().permute(0, 2, 3, 1).reshape(B, -1, 4)
            k = ks[lvl]
            if fused:
                ranked, idx = top_val[:, koff:koff + k], top_idx[:, koff:koff + k]
                koff += k
            else:
                s = cls_scores[lvl].detach().permute(0, 2, 3, 1).reshape(B, -1).sigmoid()
                ranked, idx = s.sort(dim=1, descending=True, stable=True)
                ranked, idx = ranked[:, :k], idx[:, :k]
            scores_l.append(ranked)
            deltas_l.append(torch.gather(d, 1, idx[..., None].expand(B, k, 4)))
            anchors_l.append(mlvl_anchors[lvl][idx])
            seg_sizes.append(k)
            if record:
                flat_ids.append(idx + level_off)
                level_off += Ns[lvl]
        one = len(scores_l) == 1
        scores = scores_l[0] if one else torch.cat(scores_l, 1)     # (B, K): level-major, descending inside a level
        K = scores.size(1)
        deltas = (deltas_l[0] if one else torch.cat(deltas_l, 1)).reshape(B * K, 4)
        anchors = (anchors_l[0] if one else torch.cat(anchors_l, 1)).reshape(B * K, 4)
        lim = const_tensor([[m['img_shape'][1], m['img_shape'][0]] for m in img_metas], dev, torch.float32) \
            if self.bbox_coder.clip_border else None                                      # (B, 2) w,h
        proposals = delta2bbox_clip_device(anchors, deltas, self.bbox_coder.means, self.bbox_coder.stds, lim, None,
                                           K).view(B, K, 4)
        valid = None
        if cfg.min_bbox_size > 0:
            w = proposals[..., 2] - proposals[..., 0]
            h = proposals[..., 3] - proposals[..., 1]
            valid = (w >= cfg.min_bbox_size) & (h >= cfg.min_bbox_size)
            if bool(valid.all()):
                valid = None
        if valid is not None:
            # rare path (min_bbox_size is 0 in every HTD config): fall back to the per-image operator
            from ..mmcv_ops import batched_nms
            ids = torch.cat([scores.new_full((k, ), i, dtype=torch.long) for i, k in enumerate(seg_sizes)])
            out = []
            for b in range(B):
                v = valid[b]
                dets, _ = batched_nms(proposals[b][v], scores[b][v], ids[v], dict(type='nms', iou_threshold=cfg.nms_thr))
                out.append(dets[:cfg.nms_post])
            if padded:
                n_keep = const_tensor([int(d.size(0)) for d in out], dev, torch.int64)
                return torch.stack([torch.nn.functional.pad(d, (0, 0, 0, cfg.nms_post - d.size(0))) for d in out]), n_keep
            return out
        # level id as class: shift by id*(max+1) like batched_nms does (per image), one segment per (image, level)
        ids = lvl_ids if flat_ok else torch.cat([scores.new_full((k, ), i) for i, k in enumerate(seg_sizes)])
        max_coord = proposals.reshape(B, -1).max(dim=1)[0]
        shifted = proposals + (ids.view(1, K) * (max_coord.view(B, 1) + 1)).unsqueeze(-1)
        offs = [0]
        for b in range(B):
            for k in seg_sizes:
                offs.append(offs[-1] + k)
        seg = const_tensor(offs, dev, torch.int64)
        keep = nms_sorted_mask(shifted.reshape(B * K, 4), cfg.nms_thr, 0, seg, max(seg_sizes)).view(B, K).bool()
        # survivors in descending score order (ties: lower level / lower rank first), first nms_post of them
        masked = torch.where(keep, scores, scores.new_full((1, ), -1.0))
        n_keep = keep.sum(1).clamp(max=cfg.nms_post)
        kpost = min(cfg.nms_post, K)
        if fused and kpost <= M.TOPK_KMAX:
            order, top = M.segmented_topk(masked, [(b * K, K, kpost) for b in range(B)])
            order, top = order.view(B, kpost), top.view(B, kpost)
        else:
            top, order = masked.sort(dim=1, descending=True, stable=True)
            order = order[:, :cfg.nms_post]
        if record:
            # rows of the level-concatenated candidate list in kept order (= `keep` of rpn_head.py:166-168) and the
            # anchor each of them is (flat index over the levels); entries past n_keep[b] are meaningless
            self._last_proposal_trail = (order, torch.gather(torch.cat(flat_ids, 1), 1, order), n_keep)
            self._last_candidates = (proposals, scores, ids)        # what the NMS stage was fed: (B,K,4), (B,K), (K,)
            self._last_candidate_anchors = torch.cat(flat_ids, 1)   # (B,K): flat anchor index of every candidate
        boxes = torch.gather(proposals, 1, order[..., None].expand(-1, -1, 4))
        dets = torch.cat([boxes, top[:, :cfg.nms_post, None]], -1)
        if padded:
            live = arange_cached(dets.size(1), dev)[None, :] < n_keep[:, None]
            if dets.size(1) < cfg.nms_post:
                dets = torch.nn.functional.pad(dets, (0, 0, 0, cfg.nms_post - dets.size(1)))
                live = torch.nn.functional.pad(live, (0, cfg.nms_post - live.size(1)))
            return dets * live[..., None].to(dets.dtype), n_keep
        n_keep = n_keep.tolist()
        return [dets[b, :n_keep[b]] for b in range(B)]
