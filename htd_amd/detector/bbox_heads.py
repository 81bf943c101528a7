"""Stage-1 box head and the shared BBoxHead machinery (targets, loss, refine, get_bboxes).

Reference: roi_heads/bbox_heads/bbox_head.py:13-335 (BBoxHead), convfc_bbox_head.py:8-189
(ConvFCBBoxHead / Shared2FCBBoxHead).  Registry names, kwargs and state_dict keys are the reference's.
RoI features arrive NHWC; the first FC consumes them through a (h,w,c)-ordered view of its weight, so
no activation transpose is materialised.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair

from .. import dense
from ..core import multi_apply, multiclass_nms
from ..core.misc import const_tensor
from ..registry import HEADS, build_bbox_coder, build_loss
from .bricks import ConvModule
from .losses import accuracy


class TileLinear(nn.Linear):
    """The first fully connected layer of a RoI head (convfc_bbox_head.py:118-129, htd_bbox_head.py:53): a Linear over
    the flattened (C,h,w) RoI tile.  Logically (and in checkpoints: key, shape, element order) it is the reference's
    (out, C*h*w) matrix; physically the parameter is stored (out, h, w, C) -- a 4-D channels_last tensor -- which is
    the order the NHWC RoI tiles arrive in, so neither the 51 MB weight nor its gradient is ever permuted."""

    def __init__(self, channels, tile, out_features, bias=True):
        h, w = tile
        super().__init__(channels * h * w, out_features, bias)
        self.tile_shape = (channels, h, w)
        self.weight = nn.Parameter(self.weight.data.view(out_features, channels, h, w).contiguous(
            memory_format=torch.channels_last))

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        return self

    def weight_hwc(self):
        """(out, h*w*C) zero-copy view in physical order."""
        w = self.weight
        assert w.is_contiguous(memory_format=torch.channels_last)
        return w.permute(0, 2, 3, 1).reshape(w.size(0), -1)

    def forward(self, x):
        """Reference-order 2-D input (n, C*h*w): the logical matrix on the MFMA GEMM (the heads themselves go through
        fc_on_roi_tiles and never re-order the activations).  GPU only, like every operator of this package."""
        if not x.is_cuda:
            raise NotImplementedError('TileLinear: only GPU tensors are supported (libhtd_amd.so has no CPU path)')
        from .. import dense
        return dense.linear(x, self.weight.reshape(self.out_features, -1).contiguous(), self.bias)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        destination[prefix + 'weight'] = destination[prefix + 'weight'].reshape(self.out_features, -1)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + 'weight'
        if key in state_dict and state_dict[key].dim() == 2:
            state_dict = dict(state_dict)
            state_dict[key] = state_dict[key].view(self.out_features, *self.tile_shape)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def xavier_uniform_(self):
        w2 = torch.empty(self.out_features, self.in_features, device=self.weight.device)
        nn.init.xavier_uniform_(w2)
        with torch.no_grad():
            self.weight.copy_(w2.view(self.out_features, *self.tile_shape))


def fc_on_roi_tiles(x, fc, relu=True, compute_dtype=None):
    """Linear over flattened (n,C,h,w) RoI features held NHWC.  The reference flattens in (c,h,w)
    order (convfc_bbox_head.py:147); here x is read in its physical (h,w,c) order and the weight is
    viewed in the matching order (free for a TileLinear, a permuted copy for a plain nn.Linear)."""
    n, C, h, w = x.shape
    xf = x.permute(0, 2, 3, 1).reshape(n, h * w * C)
    if compute_dtype is not None and xf.dtype != compute_dtype:
        xf = xf.to(compute_dtype)                 # bf16 configurations: the FC stack runs on the bf16 kernels
    if isinstance(fc, TileLinear):
        wt = fc.weight_hwc()
    else:
        wt = fc.weight.view(-1, C, h * w).transpose(1, 2).reshape(-1, h * w * C)
    return dense.linear(xf, wt, fc.bias, relu)


@HEADS.register_module()
class BBoxHead(nn.Module):
    def __init__(self, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7, in_channels=256,
                 num_classes=80,
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=[0., 0., 0., 0.],
                                 target_stds=[0.1, 0.1, 0.2, 0.2]),
                 reg_class_agnostic=False, reg_decoded_bbox=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)):
        super().__init__()
        assert with_cls or with_reg
        self.with_avg_pool, self.with_cls, self.with_reg = with_avg_pool, with_cls, with_reg
        self.roi_feat_size = _pair(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels, self.num_classes = in_channels, num_classes
        self.reg_class_agnostic, self.reg_decoded_bbox = reg_class_agnostic, reg_decoded_bbox
        self.fp16_enabled = False
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        in_channels = self.in_channels
        if self.with_avg_pool:
            self.avg_pool = nn.AvgPool2d(self.roi_feat_size)
        else:
            in_channels *= self.roi_feat_area
        if self.with_cls:
            self.fc_cls = nn.Linear(in_channels, num_classes + 1)
        if self.with_reg:
            self.fc_reg = nn.Linear(in_channels, 4 if reg_class_agnostic else 4 * num_classes)

    def init_weights(self):
        if self.with_cls:
            nn.init.normal_(self.fc_cls.weight, 0, 0.01)
            nn.init.constant_(self.fc_cls.bias, 0)
        if self.with_reg:
            nn.init.normal_(self.fc_reg.weight, 0, 0.001)
            nn.init.constant_(self.fc_reg.bias, 0)

    # ---------------------------------------------------------------- targets
    def _get_target_single(self, pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg):
        num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
        n = num_pos + num_neg
        labels = pos_bboxes.new_full((n, ), self.num_classes, dtype=torch.long)
        label_weights = pos_bboxes.new_zeros(n)
        bbox_targets = pos_bboxes.new_zeros(n, 4)
        bbox_weights = pos_bboxes.new_zeros(n, 4)
        if num_pos > 0:
            labels[:num_pos] = pos_gt_labels
            label_weights[:num_pos] = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
            bbox_targets[:num_pos, :] = pos_gt_bboxes if self.reg_decoded_bbox else \
                self.bbox_coder.encode(pos_bboxes, pos_gt_bboxes)
            bbox_weights[:num_pos, :] = 1
        if num_neg > 0:
            label_weights[-num_neg:] = 1.0
        return labels, label_weights, bbox_targets, bbox_weights

    def get_targets(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, concat=True):
        out = multi_apply(self._get_target_single, [r.pos_bboxes for r in sampling_results],
                          [r.neg_bboxes for r in sampling_results], [r.pos_gt_bboxes for r in sampling_results],
                          [r.pos_gt_labels for r in sampling_results], cfg=rcnn_train_cfg)
        return tuple(torch.cat(o, 0) for o in out) if concat else out

    # ---------------------------------------------------------------- loss
    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights,
             reduction_override=None, num_samples=None):
        """num_samples (device scalar, optional): number of real rows when the batch carries unused sample slots
        (static-shape training path); rows with label_weight 0 are then excluded from `acc` as well."""
        if self._fused_loss_ok(cls_score, bbox_pred, reduction_override, num_samples):
            loss_cls, acc, loss_bbox = _RoIHeadLoss.apply(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights,
                                                          num_samples, self.num_classes, float(self.loss_bbox.beta),
                                                          float(self.loss_cls.loss_weight), float(self.loss_bbox.loss_weight))
            return dict(loss_cls=loss_cls, acc=acc, loss_bbox=loss_bbox)
        losses = dict()
        if cls_score is not None:
            # avg_factor stays on the device (the reference calls .item() here: one host sync per stage)
            avg_factor = torch.sum(label_weights > 0).float().clamp(min=1.)
            if cls_score.numel() > 0:
                losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights, avg_factor=avg_factor,
                                                   reduction_override=reduction_override)
                if num_samples is None:
                    losses['acc'] = accuracy(cls_score, labels)
                else:
                    hit = (cls_score.argmax(1) == labels) & (label_weights > 0)
                    losses['acc'] = (hit.sum().float() * 100.0 / num_samples.float().clamp(min=1)).view(1)
        if bbox_pred is not None:
            pos = (labels >= 0) & (labels < self.num_classes)
            # masked form of bbox_head.py:165-185: rows of non-positives get weight 0, so no boolean gather
            # (and no .any() sync); an all-negative batch gives exactly 0 like `bbox_pred[pos_inds].sum()`
            if self.reg_decoded_bbox:
                bbox_pred = self.bbox_coder.decode(rois[:, 1:], bbox_pred)
            if self.reg_class_agnostic:
                pred = bbox_pred.view(bbox_pred.size(0), 4)
            else:
                idx = labels.clamp(max=self.num_classes - 1).view(-1, 1, 1).expand(-1, 1, 4)
                pred = torch.gather(bbox_pred.view(bbox_pred.size(0), -1, 4), 1, idx).squeeze(1)
            w = bbox_weights * pos[:, None].to(bbox_weights.dtype)
            losses['loss_bbox'] = self.loss_bbox(pred, bbox_targets, w, reduction_override=reduction_override,
                                                 avg_factor=bbox_targets.size(0) if num_samples is None else
                                                 num_samples.to(pred.dtype).clamp(min=1))
        return losses

    fused_loss = True        # one kernel for cross-entropy + smooth-L1 + accuracy on the static-shape training path

    def _fused_loss_ok(self, cls_score, bbox_pred, reduction_override, num_samples):
        """The configuration of the HTD heads (softmax cross-entropy without class weights, class-agnostic smooth-L1 on
        encoded deltas, mean reduction, static-shape batch): everything else keeps the tensor formulation below."""
        from .losses import CrossEntropyLoss, SmoothL1Loss
        lc, lb = self.loss_cls, self.loss_bbox
        return (self.fused_loss and cls_score is not None and bbox_pred is not None and num_samples is not None and
                reduction_override is None and cls_score.is_cuda and cls_score.dtype == torch.float32 and cls_score.dim() == 2 and
                0 < cls_score.size(0) and cls_score.size(1) <= 128 and bbox_pred.dtype == torch.float32 and
                self.reg_class_agnostic and not self.reg_decoded_bbox and tuple(bbox_pred.shape) == (cls_score.size(0), 4) and
                type(lc) is CrossEntropyLoss and not lc.use_sigmoid and lc.class_weight is None and lc.reduction == 'mean' and
                type(lb) is SmoothL1Loss and lb.reduction == 'mean')

    # ---------------------------------------------------------------- inference / refinement
    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        scores = F.softmax(cls_score, dim=1) if cls_score is not None else None
        if bbox_pred is not None:
            bboxes = self.bbox_coder.decode(rois[:, 1:], bbox_pred, max_shape=img_shape)
        else:
            bboxes = rois[:, 1:].clone()
            if img_shape is not None:
                bboxes[:, [0, 2]] = bboxes[:, [0, 2]].clamp(min=0, max=img_shape[1])
                bboxes[:, [1, 3]] = bboxes[:, [1, 3]].clamp(min=0, max=img_shape[0])
        if rescale and bboxes.size(0) > 0:
            if isinstance(scale_factor, float):
                bboxes = bboxes / scale_factor
            else:
                sf = const_tensor([float(v) for v in scale_factor], bboxes.device, bboxes.dtype)
                bboxes = (bboxes.view(bboxes.size(0), -1, 4) / sf).view(bboxes.size()[0], -1)
        if cfg is None:
            return bboxes, scores
        return multiclass_nms(bboxes, scores, cfg.score_thr, cfg.nms, cfg.max_per_img)

    def refine_bboxes(self, rois, labels, bbox_preds, pos_is_gts, img_metas):
        """Decode the stage's deltas onto its sampled RoIs and drop the rows that were ground truth."""
        img_ids = rois[:, 0].long()
        bboxes_list = []
        for i in range(len(img_metas)):
            inds = torch.nonzero(img_ids == i, as_tuple=False).squeeze(dim=1)
            bboxes = self.regress_by_class(rois[inds, 1:], labels[inds], bbox_preds[inds], img_metas[i])
            keep = pos_is_gts[i].new_ones(inds.numel())
            keep[:len(pos_is_gts[i])] = 1 - pos_is_gts[i]
            bboxes_list.append(bboxes[keep.type(torch.bool)])
        return bboxes_list

    def regress_by_class(self, rois, label, bbox_pred, img_meta):
        assert rois.size(1) == 4 or rois.size(1) == 5, repr(rois.shape)
        if not self.reg_class_agnostic:
            label = label * 4
            inds = torch.stack((label, label + 1, label + 2, label + 3), 1)
            bbox_pred = torch.gather(bbox_pred, 1, inds)
        assert bbox_pred.size(1) == 4
        if rois.size(1) == 4:
            return self.bbox_coder.decode(rois, bbox_pred, max_shape=img_meta['img_shape'])
        bboxes = self.bbox_coder.decode(rois[:, 1:], bbox_pred, max_shape=img_meta['img_shape'])
        return torch.cat((rois[:, [0]], bboxes), dim=1)


class _RoIHeadLoss(torch.autograd.Function):
    """BBoxHead.loss of the static-shape training path as one kernel (htd_roi_head_loss) plus a handful of scalar operations:
    -> dict(loss_cls, acc, loss_bbox) with the values of the tensor formulation (sums in another fixed order)."""

    @staticmethod
    def forward(ctx, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, num_samples, num_fg, beta, lw_cls,
                lw_box):
        from .. import capi
        n, NC = cls_score.shape
        cls = cls_score.contiguous()
        pred = bbox_pred.contiguous()
        blocks = capi.lib().htd_roi_head_loss_partial_rows()
        partial = torch.empty(blocks, 4, device=cls.device, dtype=torch.float32)
        gcls, gbox = torch.empty_like(cls), torch.empty_like(pred)
        capi.call('htd_roi_head_loss', capi.ptr(cls), capi.ptr(labels.contiguous()), capi.ptr(label_weights.float().contiguous()),
                  capi.ptr(pred), capi.ptr(bbox_targets.float().contiguous()), capi.ptr(bbox_weights.float().contiguous()), n, NC,
                  int(num_fg), float(beta), capi.ptr(partial), capi.ptr(gcls), capi.ptr(gbox), capi.current_stream_ptr())
        sums = partial.sum(0)                                     # {sum w*CE, #(w > 0), sum bw*SmoothL1, #correct}
        avg = torch.stack([sums[1], num_samples.to(torch.float32).reshape(())]).clamp(min=1.)       # avg_factor of cls, of box / acc
        scale = torch.stack([lw_cls / avg[0], lw_box / avg[1], 100.0 / avg[1]])
        vals = torch.stack([sums[0], sums[2], sums[3]]) * scale   # loss_cls, loss_bbox, acc (no index upload: that copy waits)
        loss_cls, loss_bbox, acc = vals[0], vals[1], vals[2:3]
        ctx.save_for_backward(gcls, gbox, scale)
        ctx.mark_non_differentiable(acc)
        return loss_cls, acc, loss_bbox

    @staticmethod
    def backward(ctx, g_cls, g_acc, g_box):
        gcls, gbox, scale = ctx.saved_tensors
        gc = gcls * (g_cls * scale[0]) if g_cls is not None else None
        gb = gbox * (g_box * scale[1]) if g_box is not None else None
        return gc, gb, None, None, None, None, None, None, None, None, None


@HEADS.register_module()
class ConvFCBBoxHead(BBoxHead):
    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0,
                 num_reg_fcs=0, conv_out_channels=256, fc_out_channels=1024, conv_cfg=None, norm_cfg=None, *args,
                 **kwargs):
        super().__init__(*args, **kwargs)
        assert num_shared_convs + num_shared_fcs + num_cls_convs + num_cls_fcs + num_reg_convs + num_reg_fcs > 0
        if num_cls_convs > 0 or num_reg_convs > 0:
            assert num_shared_fcs == 0
        self.num_shared_convs, self.num_shared_fcs = num_shared_convs, num_shared_fcs
        self.num_cls_convs, self.num_cls_fcs = num_cls_convs, num_cls_fcs
        self.num_reg_convs, self.num_reg_fcs = num_reg_convs, num_reg_fcs
        self.conv_out_channels, self.fc_out_channels = conv_out_channels, fc_out_channels
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.shared_convs, self.shared_fcs, last = self._add_conv_fc_branch(num_shared_convs, num_shared_fcs,
                                                                            self.in_channels, True)
        self.shared_out_channels = last
        self.cls_convs, self.cls_fcs, self.cls_last_dim = self._add_conv_fc_branch(num_cls_convs, num_cls_fcs, last)
        self.reg_convs, self.reg_fcs, self.reg_last_dim = self._add_conv_fc_branch(num_reg_convs, num_reg_fcs, last)
        if self.num_shared_fcs == 0 and not self.with_avg_pool:
            if self.num_cls_fcs == 0:
                self.cls_last_dim *= self.roi_feat_area
            if self.num_reg_fcs == 0:
                self.reg_last_dim *= self.roi_feat_area
        self.relu = nn.ReLU(inplace=True)
        if self.with_cls:
            self.fc_cls = nn.Linear(self.cls_last_dim, self.num_classes + 1)
        if self.with_reg:
            self.fc_reg = nn.Linear(self.reg_last_dim, 4 if self.reg_class_agnostic else 4 * self.num_classes)

    def _add_conv_fc_branch(self, num_branch_convs, num_branch_fcs, in_channels, is_shared=False):
        last = in_channels
        convs = nn.ModuleList()
        for i in range(num_branch_convs):
            convs.append(ConvModule(last if i == 0 else self.conv_out_channels, self.conv_out_channels, 3, padding=1,
                                    conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg))
        if num_branch_convs > 0:
            last = self.conv_out_channels
        fcs = nn.ModuleList()
        if num_branch_fcs > 0:
            if (is_shared or self.num_shared_fcs == 0) and not self.with_avg_pool:
                last *= self.roi_feat_area
            on_tiles = (is_shared or self.num_shared_fcs == 0) and not self.with_avg_pool
            for i in range(num_branch_fcs):
                if i == 0 and on_tiles:
                    fcs.append(TileLinear(last // self.roi_feat_area, self.roi_feat_size, self.fc_out_channels))
                else:
                    fcs.append(nn.Linear(last if i == 0 else self.fc_out_channels, self.fc_out_channels))
            last = self.fc_out_channels
        return convs, fcs, last

    def init_weights(self):
        super().init_weights()
        for module_list in [self.shared_fcs, self.cls_fcs, self.reg_fcs]:
            for m in module_list.modules():
                if isinstance(m, TileLinear):
                    m.xavier_uniform_()
                    nn.init.constant_(m.bias, 0)
                elif isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight)
                    nn.init.constant_(m.bias, 0)

    def _branch(self, x, convs, fcs):
        for conv in convs:
            x = conv(x)
        if x.dim() > 2:
            if self.with_avg_pool:
                from .. import mmcv_ops as M
                x = M.global_avg_pool(x).flatten(1)
            elif len(fcs) > 0:
                x, fcs = fc_on_roi_tiles(x, fcs[0], compute_dtype=getattr(self, 'compute_dtype', None)), fcs[1:]
            else:
                x = x.flatten(1)
        for fc in fcs:
            x = dense.linear(x, fc.weight, fc.bias, relu=True)
        return x.float() if x.dtype != torch.float32 else x       # classifier / regressor and the losses: fp32

    def forward(self, x):
        if self.num_shared_convs > 0 or self.num_shared_fcs > 0:
            x = self._branch(x, self.shared_convs, self.shared_fcs)
        x_cls = self._branch(x, self.cls_convs, self.cls_fcs)
        x_reg = self._branch(x, self.reg_convs, self.reg_fcs)
        cls_score = dense.linear(x_cls, self.fc_cls.weight, self.fc_cls.bias) if self.with_cls else None
        bbox_pred = dense.linear(x_reg, self.fc_reg.weight, self.fc_reg.bias) if self.with_reg else None
        return cls_score, bbox_pred


@HEADS.register_module()
class Shared2FCBBoxHead(ConvFCBBoxHead):
    def __init__(self, fc_out_channels=1024, *args, **kwargs):
        super().__init__(num_shared_convs=0, num_shared_fcs=2, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0,
                         num_reg_fcs=0, fc_out_channels=fc_out_channels, *args, **kwargs)
