"""Box arithmetic of the HTD path on the device: IoU, MaxIoU assignment, random sampling,
delta coding.  Interfaces follow the reference classes so configs build them by name:

  bbox_overlaps / BboxOverlaps2D   mmdet/core/bbox/iou_calculators/iou2d_calculator.py:43-158
  MaxIoUAssigner / AssignResult    mmdet/core/bbox/assigners/max_iou_assigner.py:10-212, assign_result.py:190-204
  RandomSampler / SamplingResult   mmdet/core/bbox/samplers/{random_sampler,base_sampler,sampling_result}.py
  DeltaXYWHBBoxCoder               mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:9-204
  bbox2roi / bbox2result           mmdet/core/bbox/transforms.py:58-116

Differences in *how* (not what): the per-gt Python loop of low-quality matching is one vectorised
max-reduction; the RNG behind sampling is pluggable (`set_randperm`) so parity tests can replay the
CPU generator the reference's CPU path would have used.
"""
import numpy as np
import torch

from ..registry import BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, IOU_CALCULATORS, build_iou_calculator
from .misc import arange_cached, const_tensor


# ------------------------------------------------------------------ IoU
def bbox_overlaps(bboxes1, bboxes2, mode='iou', is_aligned=False, eps=1e-6):
    assert mode in ('iou', 'iof'), f'Unsupported mode {mode}'
    assert bboxes1.size(-1) == 4 or bboxes1.size(0) == 0
    assert bboxes2.size(-1) == 4 or bboxes2.size(0) == 0
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new_zeros((rows, ) if is_aligned else (rows, cols))
    area1 = (bboxes1[:, 2] - bboxes1[:, 0]) * (bboxes1[:, 3] - bboxes1[:, 1])
    area2 = (bboxes2[:, 2] - bboxes2[:, 0]) * (bboxes2[:, 3] - bboxes2[:, 1])
    if is_aligned:
        wh = (torch.min(bboxes1[:, 2:], bboxes2[:, 2:]) - torch.max(bboxes1[:, :2], bboxes2[:, :2])).clamp(min=0)
        overlap = wh[:, 0] * wh[:, 1]
        union = area1 + area2 - overlap if mode == 'iou' else area1
    else:
        wh = (torch.min(bboxes1[:, None, 2:], bboxes2[None, :, 2:]) -
              torch.max(bboxes1[:, None, :2], bboxes2[None, :, :2])).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1[:, None] + area2[None, :] - overlap if mode == 'iou' else area1[:, None]
    union = torch.max(union, const_tensor([eps], union.device, union.dtype))
    return overlap / union


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:
    def __call__(self, bboxes1, bboxes2, mode='iou', is_aligned=False):
        assert bboxes1.size(-1) in [0, 4, 5] and bboxes2.size(-1) in [0, 4, 5]
        if bboxes2.size(-1) == 5:
            bboxes2 = bboxes2[..., :4]
        if bboxes1.size(-1) == 5:
            bboxes1 = bboxes1[..., :4]
        return bbox_overlaps(bboxes1, bboxes2, mode, is_aligned)

    def __repr__(self):
        return self.__class__.__name__ + '()'


# ------------------------------------------------------------------ assigner
class AssignResult:
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def add_gt_(self, gt_labels):
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long, device=gt_labels.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(len(gt_labels)), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.ignore_iof_thr = gt_max_assign_all, ignore_iof_thr
        self.ignore_wrt_candidates, self.gpu_assign_thr = ignore_wrt_candidates, gpu_assign_thr
        self.match_low_quality = match_low_quality
        self.iou_calculator = build_iou_calculator(iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0
                and bboxes.numel() > 0):
            if self.ignore_wrt_candidates:
                ignore_max, _ = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)
            else:
                ignore_max, _ = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)
            overlaps[:, ignore_max > self.ignore_iof_thr] = -1
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        if num_gts == 0 or num_bboxes == 0:
            assigned = overlaps.new_full((num_bboxes, ), 0 if num_gts == 0 else -1, dtype=torch.long)
            labels = None if gt_labels is None else overlaps.new_full((num_bboxes, ), -1, dtype=torch.long)
            return AssignResult(num_gts, assigned, overlaps.new_zeros((num_bboxes, )), labels)
        max_overlaps, argmax_overlaps = overlaps.max(dim=0)
        assigned = overlaps.new_full((num_bboxes, ), -1, dtype=torch.long)
        if isinstance(self.neg_iou_thr, float):
            neg = (max_overlaps >= 0) & (max_overlaps < self.neg_iou_thr)
        else:
            assert len(self.neg_iou_thr) == 2
            neg = (max_overlaps >= self.neg_iou_thr[0]) & (max_overlaps < self.neg_iou_thr[1])
        assigned = torch.where(neg, torch.zeros_like(assigned), assigned)
        assigned = torch.where(max_overlaps >= self.pos_iou_thr, argmax_overlaps + 1, assigned)
        if self.match_low_quality:
            # max_iou_assigner.py:184-199 loops gts in order, later gts overwrite earlier ones: for every box
            # take the LAST gt (highest index) that qualifies.
            gt_max, gt_argmax = overlaps.max(dim=1)
            ok = gt_max >= self.min_pos_iou
            ids = torch.arange(1, num_gts + 1, device=overlaps.device)
            if self.gt_max_assign_all:
                hit = (overlaps == gt_max[:, None]) & ok[:, None]
                low = (hit * ids[:, None]).max(dim=0)[0]
            else:
                # assigned[gt_argmax[i]] = i + 1 in ascending i: on duplicates the highest gt index wins
                low = torch.zeros_like(assigned).scatter_reduce(0, gt_argmax[ok], ids[ok], 'amax', include_self=True)
            assigned = torch.where(low > 0, low, assigned)
        labels = None
        if gt_labels is not None:
            pos = assigned > 0
            labels = torch.where(pos, gt_labels[(assigned - 1).clamp(min=0)], assigned.new_full((1, ), -1))
        return AssignResult(num_gts, assigned, max_overlaps, labels=labels)


# ------------------------------------------------------------------ sampler
def _device_randperm(n, device):
    return torch.randperm(n, device=device)


_randperm = _device_randperm


def set_randperm(fn=None):
    """Replace the permutation source of RandomSampler: fn(n, device) -> int64 permutation.
    None restores torch.randperm on the tensor's device (random_sampler.py:54).  Parity tests install
    `lambda n, dev: torch.randperm(n).to(dev)` to replay the CPU generator."""
    global _randperm
    _randperm = fn or _device_randperm


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            self.pos_gt_bboxes = gt_bboxes.view(-1, 4)[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction, self.neg_pos_ub = num, pos_fraction, neg_pos_ub
        self.add_gt_as_proposals = add_gt_as_proposals
        self.pos_sampler = self.neg_sampler = self

    @staticmethod
    def random_choice(gallery, num):
        assert len(gallery) >= num
        perm = _randperm(gallery.numel(), gallery.device)[:num]
        return gallery[perm]

    def _sample_pos(self, assign_result, num_expected, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(1)
        return pos_inds if pos_inds.numel() <= num_expected else self.random_choice(pos_inds, num_expected)

    def _sample_neg(self, assign_result, num_expected, **kwargs):
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(1)
        return neg_inds if len(neg_inds) <= num_expected else self.random_choice(neg_inds, num_expected)

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        if len(bboxes.shape) < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0], ), dtype=torch.uint8)
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            if gt_labels is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = torch.cat([bboxes.new_ones(gt_bboxes.shape[0], dtype=torch.uint8), gt_flags])
        num_expected_pos = int(self.num * self.pos_fraction)
        pos_inds = self._sample_pos(assign_result, num_expected_pos, bboxes=bboxes, **kwargs)
        pos_inds = torch.sort(pos_inds)[0]           # == .unique(): indices are distinct already
        num_expected_neg = self.num - pos_inds.numel()
        if self.neg_pos_ub >= 0:
            num_expected_neg = min(num_expected_neg, int(self.neg_pos_ub * max(1, pos_inds.numel())))
        neg_inds = self._sample_neg(assign_result, num_expected_neg, bboxes=bboxes, **kwargs)
        neg_inds = torch.sort(neg_inds)[0]
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


# ------------------------------------------------------------------ coder
def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    assert proposals.size() == gt.size()
    proposals, gt = proposals.float(), gt.float()
    pw = proposals[..., 2] - proposals[..., 0]
    ph = proposals[..., 3] - proposals[..., 1]
    gw = gt[..., 2] - gt[..., 0]
    gh = gt[..., 3] - gt[..., 1]
    dx = ((gt[..., 0] + gt[..., 2]) * 0.5 - (proposals[..., 0] + proposals[..., 2]) * 0.5) / pw
    dy = ((gt[..., 1] + gt[..., 3]) * 0.5 - (proposals[..., 1] + proposals[..., 3]) * 0.5) / ph
    deltas = torch.stack([dx, dy, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return deltas.sub_(const_tensor(means, deltas.device, deltas.dtype).unsqueeze(0)).div_(
        const_tensor(stds, deltas.device, deltas.dtype).unsqueeze(0))


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000, clip_border=True):
    reps = deltas.size(1) // 4
    means = const_tensor(means, deltas.device, deltas.dtype).view(1, -1).repeat(1, reps)
    stds = const_tensor(stds, deltas.device, deltas.dtype).view(1, -1).repeat(1, reps)
    d = deltas * stds + means
    dx, dy, dw, dh = d[:, 0::4], d[:, 1::4], d[:, 2::4], d[:, 3::4]
    max_ratio = np.abs(np.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    px = ((rois[:, 0] + rois[:, 2]) * 0.5).unsqueeze(1)
    py = ((rois[:, 1] + rois[:, 3]) * 0.5).unsqueeze(1)
    pw = (rois[:, 2] - rois[:, 0]).unsqueeze(1)
    ph = (rois[:, 3] - rois[:, 1]).unsqueeze(1)
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
    if clip_border and max_shape is not None:
        if isinstance(max_shape, torch.Tensor):
            # (N, 2) [h, w] per row: rows of several images in one call; min(max(x, 0), lim) is clamp's arithmetic
            mh, mw = max_shape[:, 0:1].to(x1.dtype), max_shape[:, 1:2].to(x1.dtype)
            x1, x2 = torch.minimum(x1.clamp(min=0), mw), torch.minimum(x2.clamp(min=0), mw)
            y1, y2 = torch.minimum(y1.clamp(min=0), mh), torch.minimum(y2.clamp(min=0), mh)
        else:
            x1 = x1.clamp(min=0, max=max_shape[1])
            y1 = y1.clamp(min=0, max=max_shape[0])
            x2 = x2.clamp(min=0, max=max_shape[1])
            y2 = y2.clamp(min=0, max=max_shape[0])
    return torch.stack([x1, y1, x2, y2], dim=-1).view(deltas.size())


def _f4(vals):
    import ctypes
    return (ctypes.c_float * 4)(*[float(v) for v in vals])


def delta2bbox_clip_device(rois, deltas, means, stds, lim_wh=None, keep=None, rows_per_img=None,
                           wh_ratio_clip=16 / 1000):
    """delta2bbox for (N,4) deltas + clip to per-image limits lim_wh (B,2) [w,h] (row // rows_per_img) + zeroing of
    rows with keep == False, one launch (htd_delta2bbox_clip)."""
    from .. import capi
    n = rois.size(0)
    rois, deltas = rois.float().contiguous(), deltas.float().contiguous()
    out = torch.empty(n, 4, device=rois.device, dtype=torch.float32)
    lim = lim_wh.float().contiguous() if lim_wh is not None else None
    kp = keep.to(torch.uint8).contiguous() if keep is not None else None
    capi.call('htd_delta2bbox_clip', capi.ptr(rois), capi.ptr(deltas), capi.ptr(lim), capi.ptr(kp), n,
              int(rows_per_img or max(n, 1)), _f4(means), _f4(stds), float(wh_ratio_clip), capi.ptr(out),
              capi.current_stream_ptr())
    return out


def roi_targets_device(boxes, gt_boxes, gt_labels, is_pos, valid, num_classes, means, stds):
    """-> labels (N,), label_weights (N,), bbox_targets (N,4), bbox_weights (N,4) in one launch (htd_roi_targets)."""
    from .. import capi
    n = boxes.size(0)
    dev = boxes.device
    labels = torch.empty(n, dtype=torch.int64, device=dev)
    lw = torch.empty(n, dtype=torch.float32, device=dev)
    bt = torch.empty(n, 4, dtype=torch.float32, device=dev)
    bw = torch.empty(n, 4, dtype=torch.float32, device=dev)
    # converted operands stay referenced until the launch is queued (a freed temporary's block would be reused)
    boxes, gt_boxes = boxes.float().contiguous(), gt_boxes.float().contiguous()
    gt_labels = gt_labels.to(torch.int64).contiguous()
    pos8, val8 = is_pos.to(torch.uint8).contiguous(), valid.to(torch.uint8).contiguous()
    capi.call('htd_roi_targets', capi.ptr(boxes), capi.ptr(gt_boxes), capi.ptr(gt_labels), capi.ptr(pos8),
              capi.ptr(val8), n, int(num_classes), _f4(means), _f4(stds), capi.ptr(labels), capi.ptr(lw), capi.ptr(bt),
              capi.ptr(bw), capi.current_stream_ptr())
    return labels, lw, bt, bw


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:
    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), clip_border=True):
        self.means, self.stds, self.clip_border = target_means, target_stds, clip_border

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 4
        return bbox2delta(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        return delta2bbox(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip, self.clip_border)


# ------------------------------------------------------------------ transforms
def bbox_flip(bboxes, img_shape, direction='horizontal'):
    """core/bbox/transforms.py:6-31 (tensor form of RandomFlip.bbox_flip)."""
    assert bboxes.shape[-1] % 4 == 0
    if direction not in ('horizontal', 'vertical', 'diagonal'):
        raise ValueError(f"Invalid flipping direction '{direction}'")
    flipped = bboxes.clone()
    if direction != 'vertical':
        flipped[..., 0::4] = img_shape[1] - bboxes[..., 2::4]
        flipped[..., 2::4] = img_shape[1] - bboxes[..., 0::4]
    if direction != 'horizontal':
        flipped[..., 1::4] = img_shape[0] - bboxes[..., 3::4]
        flipped[..., 3::4] = img_shape[0] - bboxes[..., 1::4]
    return flipped


def _scale_tensor(scale_factor, like):
    vals = [float(scale_factor)] if isinstance(scale_factor, (int, float)) else [float(v) for v in scale_factor]
    return const_tensor(vals, like.device, like.dtype)


def bbox_mapping(bboxes, img_shape, scale_factor, flip, flip_direction='horizontal'):
    """Original image scale -> one test-time augmentation (core/bbox/transforms.py:34-43)."""
    new_bboxes = bboxes * _scale_tensor(scale_factor, bboxes)
    return bbox_flip(new_bboxes, img_shape, flip_direction) if flip else new_bboxes


def bbox_mapping_back(bboxes, img_shape, scale_factor, flip, flip_direction='horizontal'):
    """One test-time augmentation -> original image scale (core/bbox/transforms.py:46-55)."""
    new_bboxes = bbox_flip(bboxes, img_shape, flip_direction) if flip else bboxes
    new_bboxes = new_bboxes.view(-1, 4) / _scale_tensor(scale_factor, bboxes)
    return new_bboxes.view(bboxes.shape)


def bbox2roi(bbox_list):
    rois_list = []
    for img_id, bboxes in enumerate(bbox_list):
        if bboxes.size(0) > 0:
            rois_list.append(torch.cat([bboxes.new_full((bboxes.size(0), 1), img_id), bboxes[:, :4]], dim=-1))
        else:
            rois_list.append(bboxes.new_zeros((0, 5)))
    return torch.cat(rois_list, 0)


def bbox2result_many(bboxes_list, labels_list, num_classes):
    """bbox2result of every image of a batch with ONE device-to-host copy (the per-image form copies twice per image)."""
    counts = [int(b.shape[0]) for b in bboxes_list]
    if not counts or not isinstance(bboxes_list[0], torch.Tensor) or sum(counts) == 0:
        return [bbox2result(b, l, num_classes) for b, l in zip(bboxes_list, labels_list)]
    packed = torch.cat([torch.cat(bboxes_list), torch.cat(labels_list).to(bboxes_list[0].dtype)[:, None]], 1)
    host = packed.detach().cpu().numpy()                    # labels < 2**24 are exact in fp32
    out, at = [], 0
    for c in counts:
        b, l = host[at:at + c, :5], host[at:at + c, 5].astype(np.int64)
        at += c
        out.append([np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)] if c == 0
                   else [b[l == i, :] for i in range(num_classes)])
    return out


def bbox2result(bboxes, labels, num_classes):
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


# ------------------------------------------------------------------ batched, host-sync-free forms
def _max_iou_assign_device(a, boxes, box_valid, gts, gt_valid):
    """htd_max_iou_assign (csrc/box_ops.hip): the whole assignment in one launch (two with low-quality matching)."""
    from .. import capi
    B, K = gt_valid.shape
    A = boxes.size(-2)
    shared = boxes.dim() == 2
    boxes = boxes.float().contiguous()
    gts = gts.float().contiguous()
    bv = box_valid.to(torch.uint8).contiguous()
    gv = gt_valid.to(torch.uint8).contiguous()
    assigned = torch.empty(B, A, dtype=torch.int64, device=boxes.device)
    max_ov = torch.empty(B, A, dtype=torch.float32, device=boxes.device)
    ws = torch.empty(capi.lib().htd_max_iou_assign_workspace_bytes(B, A, K) // 4, dtype=torch.float32,
                     device=boxes.device) if a.match_low_quality else None
    capi.call('htd_max_iou_assign', capi.ptr(boxes), int(shared), capi.ptr(bv), capi.ptr(gts), capi.ptr(gv), B, A, K,
              float(a.pos_iou_thr), float(a.neg_iou_thr), float(a.min_pos_iou), int(bool(a.match_low_quality)),
              capi.ptr(assigned), capi.ptr(max_ov), capi.ptr(ws), capi.current_stream_ptr())
    return assigned, max_ov


def batched_max_iou_assign(assigner, boxes, box_valid, gts, gt_valid):
    """MaxIoUAssigner.assign_wrt_overlaps (max_iou_assigner.py:124-212) for B images at once.
    boxes (A,4) shared by all images or (B,A,4); box_valid (B,A) bool; gts (B,K,4) zero-padded; gt_valid (B,K).
    -> (assigned (B,A) int64: -1 ignore / invalid box, 0 negative, k+1 matched to gt k;  max_overlaps (B,A))."""
    a = assigner
    assert isinstance(a.neg_iou_thr, float) and a.ignore_iof_thr <= 0
    if boxes.is_cuda and (a.gt_max_assign_all or not a.match_low_quality):
        return _max_iou_assign_device(a, boxes, box_valid, gts, gt_valid)
    return _batched_max_iou_assign_tensor(a, boxes, box_valid, gts, gt_valid)


def _batched_max_iou_assign_tensor(a, boxes, box_valid, gts, gt_valid):
    """The same assignment written with tensor ops (host-side logic tests on CPU tensors; the rare
    gt_max_assign_all=False configuration)."""
    if boxes.dim() == 2:
        boxes = boxes[None]
    B, K = gt_valid.shape
    area_b = (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])          # (1|B, A)
    area_g = (gts[..., 2] - gts[..., 0]) * (gts[..., 3] - gts[..., 1])
    wh = (torch.min(gts[:, :, None, 2:], boxes[:, None, :, 2:]) - torch.max(gts[:, :, None, :2], boxes[:, None, :, :2])).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = torch.max(area_g[:, :, None] + area_b[:, None, :] - overlap, const_tensor([1e-6], overlap.device, overlap.dtype))
    iou = overlap / union                                                  # (B,K,A) == bbox_overlaps(gt, boxes)
    pair_ok = gt_valid[:, :, None] & box_valid[:, None, :]
    iou = torch.where(pair_ok, iou, iou.new_full((1, ), -1.0))             # padded gts / invalid boxes never win
    max_ov, argmax = iou.max(dim=1)                                        # (B,A)
    assigned = torch.full_like(argmax, -1)
    assigned = torch.where((max_ov >= 0) & (max_ov < a.neg_iou_thr), torch.zeros_like(assigned), assigned)
    assigned = torch.where(max_ov >= a.pos_iou_thr, argmax + 1, assigned)
    if a.match_low_quality:
        gt_max, gt_arg = iou.max(dim=2)                                    # (B,K)
        ok = gt_valid & (gt_max >= a.min_pos_iou)
        ids = torch.arange(1, K + 1, device=iou.device)
        if a.gt_max_assign_all:
            hit = (iou == gt_max[:, :, None]) & ok[:, :, None] & pair_ok
            low = (hit * ids.view(1, K, 1)).max(dim=1)[0]                  # the last qualifying gt wins (:193-199)
        else:
            low = torch.zeros_like(assigned).scatter_reduce(1, gt_arg, (ids.view(1, K) * ok).expand(B, K), 'amax')
        assigned = torch.where(low > 0, low, assigned)
    has_gt = gt_valid.any(dim=1, keepdim=True)
    assigned = torch.where(has_gt, assigned, torch.zeros_like(assigned))   # image without gt: everything negative
    assigned = torch.where(box_valid, assigned, torch.full_like(assigned, -1))
    return assigned, torch.where(box_valid & has_gt, max_ov.clamp(min=0), torch.zeros_like(max_ov))


_key_source = None


def set_sample_keys(fn=None):
    """Source of the i.i.d. keys behind the batched samplers: fn(candidate_boxes (B,A,4)) -> (B,A) floats in [0,1).
    None = torch.rand on the device.  Tests install a function of the box coordinates so that differently padded
    layouts of the same candidates draw the same sample."""
    global _key_source
    _key_source = fn


def sample_keys(cand):
    if _key_source is not None:
        return _key_source(cand)
    return torch.rand(cand.shape[:2], device=cand.device)


_SAMPLE_PLANS = {}      # (B, A, kpos, kneg, device) -> device tables of htd_random_sample


def random_sample_device(assigned, keys, num, pos_fraction, neg_pos_ub=-1, slots=0):
    """RandomSampler for a batch in one C-ABI call (htd_random_sample: key staging, segmented top-k, masks).
    -> pos_mask, neg_mask (B,A) bool, counts (B,2) drawn (pos, neg), order (B,slots) or None."""
    from .. import capi
    from ..mmcv_ops import TOPK_CHUNK
    B, A = assigned.shape
    dev = assigned.device
    max_pos = int(num * pos_fraction)
    kpos, kneg = min(max_pos, A), min(num, A)
    key = (B, A, kpos, kneg, str(dev))
    plan = _SAMPLE_PLANS.get(key)
    if plan is None:
        rows = [(b * A, A, kpos, b * kpos) for b in range(B)] + [(B * A + b * A, A, kneg, B * kpos + b * kneg) for b in range(B)]
        per = (A + TOPK_CHUNK - 1) // TOPK_CHUNK
        chunks = [(s, c) for s in range(2 * B) for c in range(per)]
        if len(_SAMPLE_PLANS) > 64:
            _SAMPLE_PLANS.clear()
        plan = _SAMPLE_PLANS[key] = (torch.tensor(rows, dtype=torch.int64, device=dev),
                                     torch.tensor(chunks, dtype=torch.int32, device=dev), len(chunks))
    segs, tab, nchunks = plan
    pos = torch.empty(B, A, dtype=torch.bool, device=dev)
    neg = torch.empty(B, A, dtype=torch.bool, device=dev)
    counts = torch.empty(B, 2, dtype=torch.int64, device=dev)
    order = torch.empty(B, slots, dtype=torch.int64, device=dev) if slots > 0 else None
    ws = torch.empty(capi.lib().htd_random_sample_workspace_bytes(B, A, nchunks), dtype=torch.uint8, device=dev)
    capi.call('htd_random_sample', capi.ptr(assigned.contiguous()), capi.ptr(keys.contiguous()), B, A, int(num), max_pos,
              float(neg_pos_ub), capi.ptr(segs), capi.ptr(tab), nchunks, capi.ptr(pos), capi.ptr(neg), capi.ptr(counts),
              capi.ptr(order) if order is not None else None, int(slots), capi.ptr(ws), capi.current_stream_ptr())
    return pos, neg, counts, order


def _sample_on_device(assigned, keys, num):
    return assigned.is_cuda and assigned.dtype == torch.int64 and keys.dtype == torch.float32 and 0 < num <= 2048


def batched_random_sample(assigned, num, pos_fraction, neg_pos_ub=-1, keys=None):
    """RandomSampler (base_sampler.py:34-101, random_sampler.py:58-78) without host round trips: a uniformly random
    subset of size n is the n candidates with the smallest i.i.d. random keys.  -> (pos_mask, neg_mask) (B,A)."""
    B, A = assigned.shape
    if keys is None:
        keys = torch.rand(B, A, device=assigned.device)
    if _sample_on_device(assigned, keys, num):
        return random_sample_device(assigned, keys, num, pos_fraction, neg_pos_ub)[:2]
    is_pos, is_neg = assigned > 0, assigned == 0
    ar = torch.arange(A, device=assigned.device).expand(B, A)

    def pick(cand, limit):
        order = torch.where(cand, keys, keys.new_full((1, ), 2.0)).argsort(dim=1, stable=True)
        rank = torch.empty_like(order).scatter_(1, order, ar)
        return cand & (rank < limit)
    pos_mask = pick(is_pos, torch.full((B, 1), int(num * pos_fraction), device=assigned.device))
    n_pos = pos_mask.sum(dim=1, keepdim=True)
    n_neg = num - n_pos
    if neg_pos_ub >= 0:
        n_neg = torch.min(n_neg, (neg_pos_ub * n_pos.clamp(min=1)).long())
    return pos_mask, pick(is_neg, n_neg)


class BatchSamplingResult:
    """Per-image view with the attributes of SamplingResult (sampling_result.py:6-60), cut out of batched tensors."""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, gt_inds, labels, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes.view(-1, 4)[self.pos_assigned_gt_inds, :] if gt_bboxes.numel() else \
            torch.empty_like(gt_bboxes).view(-1, 4)
        self.pos_gt_labels = labels[pos_inds] if labels is not None else None

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


def batched_assign_and_sample(assigner, sampler, proposal_list, gt_bboxes, gt_labels, keys=None):
    """assign + sample of every image of the batch (htd_roi_head.py:254-264,292-310) with ONE device->host copy
    (the sample counts).  Candidate order per image is [gt boxes, proposals] when add_gt_as_proposals
    (base_sampler.py:72-81); sampled indices are ascending like `.unique()` leaves them.
    -> (list of BatchSamplingResult, counts) with counts[b] = (n_pos, n_neg, n_pos_that_are_gt)."""
    dev = proposal_list[0].device
    B = len(proposal_list)
    add_gt = sampler.add_gt_as_proposals
    P = max(1, max(int(p.size(0)) for p in proposal_list))
    K = max(1, max(int(g.size(0)) for g in gt_bboxes))
    props = proposal_list[0].new_zeros(B, P, 4)
    pvalid = torch.zeros(B, P, dtype=torch.bool, device=dev)
    gts = proposal_list[0].new_zeros(B, K, 4)
    gvalid = torch.zeros(B, K, dtype=torch.bool, device=dev)
    glabels = torch.zeros(B, K, dtype=torch.long, device=dev)
    for b in range(B):
        n, k = proposal_list[b].size(0), gt_bboxes[b].size(0)
        if n:
            props[b, :n] = proposal_list[b][:, :4]
            pvalid[b, :n] = True
        if k:
            gts[b, :k] = gt_bboxes[b][:, :4]
            gvalid[b, :k] = True
            glabels[b, :k] = gt_labels[b]
    assigned, _ = batched_max_iou_assign(assigner, props, pvalid, gts, gvalid)
    labels = torch.where(assigned > 0, torch.gather(glabels, 1, (assigned - 1).clamp(min=0)), torch.full_like(assigned, -1))
    if add_gt:      # AssignResult.add_gt_: gt i is a candidate matched to itself
        self_inds = torch.where(gvalid, arange_cached(K, dev, start=1).expand(B, K), const_tensor([-1], dev, torch.int64))
        assigned = torch.cat([self_inds, assigned], 1)
        labels = torch.cat([torch.where(gvalid, glabels, torch.full_like(glabels, -1)), labels], 1)
        cand = torch.cat([gts, props], 1)
        is_gt = torch.cat([gvalid, torch.zeros_like(pvalid)], 1)
    else:
        cand, is_gt = props, torch.zeros_like(pvalid)
    pos, neg = batched_random_sample(assigned, sampler.num, sampler.pos_fraction, sampler.neg_pos_ub,
                                     keys if keys is not None else sample_keys(cand))
    # selected candidates first, ascending index inside (stable sort of the complement mask)
    pos_order = torch.sort((~pos).to(torch.uint8), dim=1, stable=True)[1]
    neg_order = torch.sort((~neg).to(torch.uint8), dim=1, stable=True)[1]
    counts = torch.stack([pos.sum(1), neg.sum(1), (pos & is_gt).sum(1)], 1).tolist()        # the one host read
    out = []
    gt_flags = is_gt.to(torch.uint8)
    for b in range(B):
        npos, nneg, _ = counts[b]
        k = gt_bboxes[b].size(0)
        # indices refer to the padded candidate layout; remap to the reference's compact [gt_b ; proposals_b] layout
        pi, ni = pos_order[b, :npos], neg_order[b, :nneg]
        if add_gt and k < K:
            shift = K - k
            pi = torch.where(pi >= K, pi - shift, pi)
            ni = torch.where(ni >= K, ni - shift, ni)
            boxes_b = torch.cat([gt_bboxes[b][:, :4], proposal_list[b][:, :4]], 0)
            inds_b = torch.cat([assigned[b, :k], assigned[b, K:K + proposal_list[b].size(0)]])
            lab_b = torch.cat([labels[b, :k], labels[b, K:K + proposal_list[b].size(0)]])
            flg_b = torch.cat([gt_flags[b, :k], gt_flags[b, K:K + proposal_list[b].size(0)]])
        else:
            n_b = (K if add_gt else 0) + proposal_list[b].size(0)
            boxes_b, inds_b, lab_b, flg_b = cand[b, :n_b], assigned[b, :n_b], labels[b, :n_b], gt_flags[b, :n_b]
        out.append(BatchSamplingResult(pi, ni, boxes_b, gt_bboxes[b][:, :4], inds_b, lab_b, flg_b))
        out[-1].num_pos_gt = counts[b][2]           # host copy of pos_is_gt.sum() (saves refine_bboxes a device read)
    return out, counts


_PAD_GT_LAST = {}
_PAD_GT_CACHE = __import__('os').environ.get('HTD_PAD_GT_CACHE', '1') != '0'


def pad_gt_batch(gt_bboxes, gt_labels=None):
    """Per-image gt lists -> zero-padded (B,K,4) boxes, (B,K) validity [, (B,K) labels] with two launches instead of
    three slice assignments per image: one cat, one gather through an index built from the (host-known) list sizes.
    The RPN targets and both RoI stages pad the same lists: the last result is kept (keyed by the tensors' identity and
    version; it holds them alive, so an address cannot be recycled under the key)."""
    key = (tuple((g.data_ptr(), g._version, tuple(g.shape)) for g in gt_bboxes),
           None if gt_labels is None else tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in gt_labels))
    hit = _PAD_GT_LAST.get(key) if _PAD_GT_CACHE else None
    if hit is not None:
        return hit[0]
    out = _pad_gt_batch(gt_bboxes, gt_labels)
    if len(_PAD_GT_LAST) >= 4:
        _PAD_GT_LAST.clear()
    _PAD_GT_LAST[key] = (out, list(gt_bboxes), None if gt_labels is None else list(gt_labels))
    return out


def _pad_gt_batch(gt_bboxes, gt_labels=None):
    dev = gt_bboxes[0].device
    sizes = tuple(int(g.size(0)) for g in gt_bboxes)
    B, K, tot = len(sizes), max(1, max(sizes)), sum(sizes)
    idx, off = [], 0
    for k in sizes:
        idx.append(list(range(off, off + k)) + [tot] * (K - k))      # `tot` = the appended all-zero row
        off += k
    idx_t = const_tensor(idx, dev, torch.int64)
    valid = const_tensor([[j < k for j in range(K)] for k in sizes], dev, torch.bool)
    flat = torch.cat([g[:, :4] for g in gt_bboxes] + [gt_bboxes[0].new_zeros(1, 4)])
    gts = flat[idx_t.view(-1)].view(B, K, 4)
    if gt_labels is None:
        return gts, valid
    lab = torch.cat(list(gt_labels) + [gt_labels[0].new_zeros(1)])
    return gts, valid, lab[idx_t.view(-1)].view(B, K).long()


class StaticSamples:
    """Sampling result of a whole batch in FIXED slots: S = sampler.num rows per image, the sampled positives first
    (ascending candidate index), then the sampled negatives -- the order of SamplingResult.bboxes
    (sampling_result.py:40-43) -- then unused slots.  Every member is a device tensor of static shape, so the RoI
    head that consumes it never has to read a count back to the host."""

    def __init__(self, boxes, valid, is_pos, npos, nneg, pos_gt_bboxes, pos_gt_labels, pos_is_gt):
        self.boxes, self.valid, self.is_pos = boxes, valid, is_pos                  # (B,S,4), (B,S), (B,S)
        self.npos, self.nneg = npos, nneg                                          # (B,)
        self.pos_gt_bboxes, self.pos_gt_labels, self.pos_is_gt = pos_gt_bboxes, pos_gt_labels, pos_is_gt

    @property
    def rois(self):
        B, S = self.valid.shape
        img = arange_cached(B, self.boxes.device, self.boxes.dtype).view(B, 1, 1).expand(B, S, 1)
        return torch.cat([img, self.boxes], -1).view(B * S, 5)


STATIC_FINISH = __import__('os').environ.get('HTD_STATIC_FINISH', '1') != '0'       # 0: the tensor formulation (A/B runs, tests)


def static_assign_and_sample(assigner, sampler, props, pvalid, gt_bboxes, gt_labels):
    """MaxIoUAssigner + RandomSampler for every image (htd_roi_head.py:254-264,292-310) with NO device->host
    copy: props (B,P,4) zero-padded with validity mask pvalid (B,P).  -> StaticSamples."""
    dev = props.device
    B, P = pvalid.shape
    S = sampler.num
    gts, gvalid, glabels = pad_gt_batch(gt_bboxes, gt_labels)
    K = gts.size(1)
    assigned, _ = batched_max_iou_assign(assigner, props, pvalid, gts, gvalid)
    add_gt = bool(sampler.add_gt_as_proposals)
    if add_gt:                           # AssignResult.add_gt_: gt i is a candidate matched to itself
        self_inds = torch.where(gvalid, arange_cached(K, dev, start=1).expand(B, K), const_tensor([-1], dev, torch.int64))
        assigned = torch.cat([self_inds, assigned], 1)
        cand = torch.cat([gts, props], 1)
    else:
        cand = props
    A = cand.size(1)
    if A < S:                            # fewer candidates than slots: pad with candidates that can never be drawn
        cand = torch.cat([cand, cand.new_zeros(B, S - A, 4)], 1)
        assigned = torch.cat([assigned, assigned.new_full((B, S - A), -1)], 1)
        A = S
    keys = sample_keys(cand)
    if _sample_on_device(assigned, keys, S) and STATIC_FINISH and props.dtype == torch.float32:
        # masks, drawn counts and the slot order (drawn positives, then drawn negatives, ascending index) from one call, the
        # fixed-slot result from a second (htd_static_samples_finish) instead of ~14 gather / compare / concatenate launches
        from .. import capi
        pos, neg, counts, order = random_sample_device(assigned, keys, S, sampler.pos_fraction, sampler.neg_pos_ub, slots=S)
        boxes = torch.empty(B, S, 4, device=dev, dtype=torch.float32)
        valid, is_pos, pos_is_gt = (torch.empty(B, S, device=dev, dtype=torch.bool) for _ in range(3))
        pgb = torch.empty(B, S, 4, device=dev, dtype=torch.float32)
        pgl = torch.empty(B, S, device=dev, dtype=torch.int64)
        P_ = props.size(1)
        capi.call('htd_static_samples_finish', capi.ptr(gts.contiguous()), capi.ptr(gvalid.contiguous()), capi.ptr(glabels.contiguous()),
                  capi.ptr(props.contiguous()), capi.ptr(assigned.contiguous()), capi.ptr(order), capi.ptr(counts), B, K, P_, A, S,
                  int(add_gt), capi.ptr(boxes), capi.ptr(valid), capi.ptr(is_pos), capi.ptr(pgb), capi.ptr(pgl), capi.ptr(pos_is_gt),
                  capi.current_stream_ptr())
        return StaticSamples(boxes, valid, is_pos, counts[:, 0], counts[:, 1], pgb, pgl, pos_is_gt)
    is_gt = torch.cat([gvalid, torch.zeros_like(pvalid)], 1) if add_gt else torch.zeros_like(pvalid)
    if is_gt.size(1) < A:
        is_gt = torch.cat([is_gt, is_gt.new_zeros(B, A - is_gt.size(1))], 1)
    if _sample_on_device(assigned, keys, S):
        # masks, drawn counts and the slot order (drawn positives, then drawn negatives, ascending index) from one call
        pos, neg, counts, order = random_sample_device(assigned, keys, S, sampler.pos_fraction, sampler.neg_pos_ub, slots=S)
        npos, nneg = counts[:, 0], counts[:, 1]
    else:
        pos, neg = batched_random_sample(assigned, S, sampler.pos_fraction, sampler.neg_pos_ub, keys)
        ar = torch.arange(A, device=dev).expand(B, A)
        order = torch.where(pos, ar, torch.where(neg, ar + A, ar + 2 * A)).argsort(dim=1)[:, :S]      # (B,S)
        npos, nneg = pos.sum(1), neg.sum(1)
    slot = arange_cached(S, dev).expand(B, S)
    valid = slot < (npos + nneg)[:, None]
    is_pos = slot < npos[:, None]
    boxes = torch.gather(cand, 1, order[..., None].expand(B, S, 4)) * valid[..., None].to(cand.dtype)
    gt_idx = (torch.gather(assigned, 1, order) - 1).clamp(min=0)
    return StaticSamples(boxes, valid, is_pos, npos, nneg,
                         torch.gather(gts, 1, gt_idx[..., None].expand(B, S, 4)), torch.gather(glabels, 1, gt_idx),
                         torch.gather(is_gt, 1, order) & is_pos)
