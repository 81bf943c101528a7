"""FasterRCNN / TwoStageDetector / BaseDetector glue of the path.

Reference: detectors/base.py:15-372 (forward :168-182, _parse_losses :184-223, train_step :225-258),
two_stage.py:10-222 (extract_feat :80-87, forward_train :107-170, simple_test :190-211),
faster_rcnn.py:5-24.  `_parse_losses` packs all log scalars into ONE all-reduce and ONE device->host
copy (the reference does ~9 of each per step, base.py:216-221).
"""
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import mmcv_ops as M
from ..registry import DETECTORS, build_backbone, build_head, build_neck


class BaseDetector(nn.Module):
    def __init__(self):
        super().__init__()
        self.fp16_enabled = False

    @property
    def with_neck(self):
        return getattr(self, 'neck', None) is not None

    @property
    def with_bbox(self):
        return getattr(self, 'roi_head', None) is not None and self.roi_head.with_bbox

    @property
    def with_rpn(self):
        return getattr(self, 'rpn_head', None) is not None

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        self._drop_step_caches(imgs[0])
        for img, img_meta in zip(imgs, img_metas):
            for m in img_meta:
                m['batch_intput_shape'] = tuple(img.size()[-2:])
        if len(imgs) == 1:
            if 'proposals' in kwargs:
                kwargs['proposals'] = kwargs['proposals'][0]
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        assert imgs[0].size(0) == 1, f'aug test does not support inference with batch size {imgs[0].size(0)}'
        assert 'proposals' not in kwargs
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def _parse_losses(self, losses):
        """detectors/base.py:196-223.  Values that already are scalars (every loss of this path) skip their `.mean()`, and the
        total is ONE stacked sum whose backward hands the incoming gradient to every term as it is: the reference's chain of
        `mean` and `+` nodes is ~25 element-wise launches on 0-dim tensors per step, forward and backward."""
        def scalar(v):
            return v.reshape(()) if v.numel() == 1 else v.mean()
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = scalar(value)
            elif isinstance(value, list):
                log_vars[name] = sum(scalar(v) for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        terms = [v for k, v in log_vars.items() if 'loss' in k]
        loss = _SumScalars.apply(*terms) if len(terms) > 1 and all(t.dim() == 0 and t.is_cuda for t in terms) else sum(terms)
        log_vars['loss'] = loss
        packed = torch.stack([v.detach().reshape(()).float() for v in log_vars.values()])
        if dist.is_available() and dist.is_initialized():
            packed = packed / dist.get_world_size()
            dist.all_reduce(packed)
        self._last_log_tensor = packed                      # rank-averaged scalars, still on the device
        self._last_log_keys = list(log_vars.keys())
        return loss, LazyLogVars(self._last_log_keys, packed)

    def train_step(self, data, optimizer):
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    val_step = train_step


class _SumScalars(torch.autograd.Function):
    """sum of 0-dim tensors, added left to right like Python's sum(); every input receives the output's gradient unchanged."""

    @staticmethod
    def forward(ctx, *terms):
        ctx.n = len(terms)
        total = terms[0] + terms[1]
        for t in terms[2:]:
            total = total + t
        return total

    @staticmethod
    def backward(ctx, g):
        return (g, ) * ctx.n


class LazyLogVars(OrderedDict):
    """log_vars whose float values are fetched with a single device->host copy on first access."""

    def __init__(self, keys, packed):
        super().__init__()
        self._keys, self._packed, self._done = keys, packed, False

    def _fill(self):
        if not self._done:
            self._done = True
            for k, v in zip(self._keys, self._packed.tolist()):
                super().__setitem__(k, v)

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def items(self):
        self._fill()
        return super().items()

    def keys(self):
        return list(self._keys)

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def values(self):
        self._fill()
        return super().values()


@DETECTORS.register_module()
class TwoStageDetector(BaseDetector):
    def __init__(self, backbone, neck=None, rpn_head=None, roi_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None):
        super().__init__()
        self.backbone = build_backbone(backbone)
        if neck is not None:
            self.neck = build_neck(neck)
        if rpn_head is not None:
            rpn_train_cfg = train_cfg.rpn if train_cfg is not None else None
            rpn_head_ = dict(rpn_head)
            rpn_head_.update(train_cfg=rpn_train_cfg, test_cfg=test_cfg.rpn if test_cfg is not None else None)
            self.rpn_head = build_head(rpn_head_)
        if roi_head is not None:
            rcnn_train_cfg = train_cfg.rcnn if train_cfg is not None else None
            roi_head_ = dict(roi_head)
            roi_head_.update(train_cfg=rcnn_train_cfg, test_cfg=test_cfg.rcnn if test_cfg is not None else None)
            self.roi_head = build_head(roi_head_)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained=pretrained)

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            self.neck.init_weights()
        if self.with_rpn:
            self.rpn_head.init_weights()
        if getattr(self, 'roi_head', None) is not None:
            self.roi_head.init_weights(pretrained)

    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    def extract_feats(self, imgs):
        """detectors/base.py:51-63: one feature pyramid per test-time augmentation."""
        assert isinstance(imgs, list)
        return [self.extract_feat(img) for img in imgs]

    def aug_test(self, imgs, img_metas, rescale=False):
        """detectors/two_stage.py:213-222: multi-scale / flip test-time augmentation of ONE image."""
        x = self.extract_feats(imgs)
        proposal_list = self.rpn_head.aug_test_rpn(x, img_metas)
        x32 = [f if f[0].dtype == torch.float32 else tuple(t.float() for t in f) for f in x]
        return self.roi_head.aug_test(x32, proposal_list, img_metas, rescale=rescale)

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None,
                      proposals=None, **kwargs):
        if img.is_cuda:
            from .. import dense
            dense.new_step()
            if torch.is_grad_enabled():
                # the dgrad images of every plain convolution / Linear of neck and heads in one launch (dense.flip_many)
                dense.flip_many(self._plain_weights())
        x = self.extract_feat(img)
        losses = dict()
        x32 = x if x[0].dtype == torch.float32 else tuple(f.float() for f in x)     # RoI head: fp32 (force_fp32 sites)
        if self.with_rpn:
            proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg.rpn)
            static = x[0].is_cuda and gt_masks is None and not kwargs and \
                getattr(self.roi_head, 'can_train_static', lambda *_: False)(gt_bboxes_ignore)
            chain = x[0].is_cuda and x[0].dtype == torch.float32 and torch.is_grad_enabled()
            feats = M.PyramidTaps(x) if chain else x
            rpn_losses, proposal_list = self.rpn_head.forward_train(feats, img_metas, gt_bboxes, gt_labels=None,
                                                                    gt_bboxes_ignore=gt_bboxes_ignore,
                                                                    proposal_cfg=proposal_cfg, padded=static)
            losses.update(rpn_losses)
            if chain:       # the RoI head reads the aliases the RPN convolutions handed back: one gradient map per level
                x = x32 = tuple(feats.levels)
            if static:       # fixed-shape RoI head: the whole train step runs without a host/device synchronisation
                losses.update(self.roi_head.forward_train_static(x32, img_metas, *proposal_list, gt_bboxes, gt_labels))
                return losses
        else:
            proposal_list = proposals
        losses.update(self.roi_head.forward_train(x32, img_metas, proposal_list, gt_bboxes, gt_labels,
                                                  gt_bboxes_ignore, gt_masks, **kwargs))
        return losses

    def _plain_weights(self):
        """Trainable fp32 weights of the convolutions / Linear layers outside the backbone (whose stages fold and flip their
        own): the operands of this step's data gradients."""
        ws = getattr(self, '_plain_weight_list', None)
        if ws is None:
            ws = []
            for part in (getattr(self, 'neck', None), getattr(self, 'rpn_head', None), getattr(self, 'roi_head', None)):
                if part is None:
                    continue
                for m in part.modules():
                    if isinstance(m, (nn.Conv2d, nn.Linear)) and getattr(m, 'groups', 1) == 1 and m.weight.requires_grad:
                        ws.append(m.weight)
            self._plain_weight_list = ws
        return ws

    @staticmethod
    def _drop_step_caches(img):
        """A test pass with autograd ENABLED folds BN into fresh weight tensors on every call (the training branch of
        frozen_bn_fold_many); their plane / flipped images are keyed by those tensors and would pile up in dense's per-step
        caches, one backbone of weights per call.  Under no_grad the folded weights are cached and so are their images."""
        if img.is_cuda and torch.is_grad_enabled():
            from .. import dense
            dense.new_step()

    def simple_test(self, img, img_metas, proposals=None, rescale=False):
        assert self.with_bbox, 'Bbox head must be implemented.'
        self._drop_step_caches(img)
        x = self.extract_feat(img)
        proposal_list = self.rpn_head.simple_test_rpn(x, img_metas) if proposals is None else proposals
        x32 = x if x[0].dtype == torch.float32 else tuple(f.float() for f in x)
        return self.roi_head.simple_test(x32, proposal_list, img_metas, rescale=rescale)


@DETECTORS.register_module()
class FasterRCNN(TwoStageDetector):
    def __init__(self, backbone, rpn_head, roi_head, train_cfg, test_cfg, neck=None, pretrained=None):
        super().__init__(backbone=backbone, neck=neck, rpn_head=rpn_head, roi_head=roi_head, train_cfg=train_cfg,
                         test_cfg=test_cfg, pretrained=pretrained)
