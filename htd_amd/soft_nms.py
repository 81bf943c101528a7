"""mmcv.ops.soft_nms / batched soft-NMS on the device (R101 test configs, configs/htd/htd_resnet101_2x.py:298)."""
import torch

from . import capi

_P, _S = capi.ptr, capi.current_stream_ptr
_METHOD = {'naive': 0, 'linear': 1, 'gaussian': 2}


def _run(boxes_for_nms, scores, seg, iou_threshold, sigma, min_score, method, offset):
    n = boxes_for_nms.size(0)
    new_scores = scores.clone().float().contiguous()
    rank = torch.empty(n, dtype=torch.int32, device=scores.device)
    boxes_f = boxes_for_nms.float().contiguous()      # referenced until the launch is queued
    capi.call('htd_soft_nms_segments', _P(boxes_f), _P(new_scores), _P(seg),
              seg.numel() - 1, n, _P(rank), float(iou_threshold), float(sigma), float(min_score), _METHOD[method],
              int(offset), _S())
    return new_scores, rank


def soft_nms(boxes, scores, iou_threshold=0.3, sigma=0.5, min_score=1e-3, method='linear', offset=0, iou_thr=None):
    """-> (dets (k,5) with decayed scores, inds (k,)) in selection order (descending decayed score)."""
    if iou_thr is not None:          # deprecated spelling still used by the reference configs
        iou_threshold = iou_thr
    if not boxes.is_cuda:
        raise NotImplementedError('soft_nms: only GPU tensors are supported')
    assert boxes.size(1) == 4 and boxes.size(0) == scores.size(0) and method in _METHOD
    n = boxes.size(0)
    if n == 0:
        return boxes.new_zeros((0, 5)), boxes.new_zeros((0, ), dtype=torch.long)
    seg = torch.tensor([0, n], dtype=torch.int64, device=boxes.device)
    new_scores, rank = _run(boxes, scores, seg, iou_threshold, sigma, min_score, method, offset)
    kept = (rank >= 0).nonzero(as_tuple=False).squeeze(1)
    inds = kept[torch.sort(rank[kept], stable=True)[1]]
    return torch.cat([boxes[inds], new_scores[inds, None]], 1), inds


def soft_nms_batched(boxes, scores, idxs, class_agnostic=False, iou_threshold=0.3, sigma=0.5, min_score=1e-3,
                     method='linear', offset=0, iou_thr=None, **unused):
    """batched_nms(..., dict(type='soft_nms', ...)): classes are independent segments of one launch; the merged
    output order is descending decayed score, as the single mmcv call over offset-shifted boxes produces."""
    if iou_thr is not None:
        iou_threshold = iou_thr
    if not boxes.is_cuda:
        raise NotImplementedError('soft_nms: only GPU tensors are supported')
    n = boxes.size(0)
    if n == 0:
        return boxes.new_zeros((0, 5)), boxes.new_zeros((0, ), dtype=torch.long)
    if class_agnostic:
        boxes_for_nms, idxs = boxes, torch.zeros_like(idxs)
    else:
        boxes_for_nms = boxes + (idxs.to(boxes) * (boxes.max() + 1))[:, None]
    perm = torch.sort(idxs, stable=True)[1]
    counts = torch.bincount(idxs[perm])
    seg = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=boxes.device)
    seg[1:] = torch.cumsum(counts, 0)
    new_sorted, rank_sorted = _run(boxes_for_nms[perm], scores[perm], seg, iou_threshold, sigma, min_score, method,
                                   offset)
    new_scores = torch.empty_like(new_sorted)
    new_scores[perm] = new_sorted
    kept_mask = torch.zeros(n, dtype=torch.bool, device=boxes.device)
    kept_mask[perm] = rank_sorted >= 0
    kept = kept_mask.nonzero(as_tuple=False).squeeze(1)
    if n >= 10000:
        # mmcv's split_thr branch keeps the ORIGINAL scores and orders by them (nms.py batched_nms, >= 10000 boxes)
        keep = kept[torch.sort(scores[kept], descending=True, stable=True)[1]]
        return torch.cat([boxes[keep], scores[keep, None]], 1), keep
    keep = kept[torch.sort(new_scores[kept], descending=True, stable=True)[1]]
    return torch.cat([boxes[keep], new_scores[keep, None]], 1), keep
