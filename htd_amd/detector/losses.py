"""Losses of the HTD path: CrossEntropyLoss (softmax / sigmoid), SmoothL1Loss, accuracy.
Reference: mmdet/models/losses/{cross_entropy_loss.py:9-202, smooth_l1_loss.py:8-94, utils.py:26-52,
accuracy.py:4-48}.  Same constructor kwargs and forward signature (weight, avg_factor,
reduction_override)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import LOSSES
from ..core.misc import const_tensor


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        return loss.sum() if reduction == 'sum' else loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    loss = F.cross_entropy(pred, label, weight=class_weight, reduction='none')
    return weight_reduce_loss(loss, None if weight is None else weight.float(), reduction, avg_factor)


def binary_cross_entropy(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    if pred.dim() != label.dim():
        # integer labels -> one-hot columns; out-of-range labels (background) give an all-zero row
        C = pred.size(-1)
        onehot = (label.view(-1, 1) == torch.arange(C, device=label.device).view(1, -1))
        label = onehot
        weight = None if weight is None else weight.view(-1, 1).expand(weight.size(0), C)
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), pos_weight=class_weight, reduction='none')
    return weight_reduce_loss(loss, None if weight is None else weight.float(), reduction, avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        assert not use_mask, 'mask cross-entropy is outside the HTD path'
        self.use_sigmoid, self.use_mask, self.reduction = use_sigmoid, use_mask, reduction
        self.loss_weight, self.class_weight = loss_weight, class_weight
        self.cls_criterion = binary_cross_entropy if use_sigmoid else cross_entropy

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        cw = const_tensor(self.class_weight, cls_score.device, cls_score.dtype) if self.class_weight is not None else None
        return self.loss_weight * self.cls_criterion(cls_score, label, weight, class_weight=cw, reduction=reduction,
                                                     avg_factor=avg_factor, **kwargs)


def smooth_l1_loss(pred, target, weight=None, beta=1.0, reduction='mean', avg_factor=None):
    assert beta > 0
    assert pred.size() == target.size() and target.numel() > 0
    diff = torch.abs(pred - target)
    loss = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta, reduction=reduction,
                                                 avg_factor=avg_factor, **kwargs)


def accuracy(pred, target, topk=1, thresh=None):
    assert isinstance(topk, (int, tuple))
    single = isinstance(topk, int)
    topk = (topk, ) if single else topk
    if pred.size(0) == 0:
        accu = [pred.new_tensor(0.) for _ in topk]
        return accu[0] if single else accu
    assert pred.ndim == 2 and target.ndim == 1 and pred.size(0) == target.size(0)
    maxk = max(topk)
    assert maxk <= pred.size(1), f'maxk {maxk} exceeds pred dimension {pred.size(1)}'
    pred_value, pred_label = pred.topk(maxk, dim=1)
    pred_label = pred_label.t()
    correct = pred_label.eq(target.view(1, -1).expand_as(pred_label))
    if thresh is not None:
        correct = correct & (pred_value > thresh).t()
    res = [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / pred.size(0)) for k in topk]
    return res[0] if single else res
