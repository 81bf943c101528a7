"""multiclass_nms (mmdet/core/post_processing/bbox_nms.py:7-71) and the test-time-augmentation merges
(mmdet/core/post_processing/merge_augs.py:9-92) on the device."""
import torch

from ..mmcv_ops import batched_nms, nms
from .bbox import bbox_mapping_back


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """multi_bboxes (n, #class*4) or (n, 4); multi_scores (n, #class+1), last column = background.
    -> (dets (k,5), labels (k,)) sorted by descending score, at most max_num."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    valid_mask = scores > score_thr
    bboxes = bboxes[valid_mask]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid_mask]
    labels = valid_mask.nonzero(as_tuple=False)[:, 1]
    if bboxes.numel() == 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0, ), dtype=torch.long)
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep]


def multiclass_nms_images(multi_bboxes, multi_scores, img_of, num_imgs, score_thr, nms_cfg, max_num=-1):
    """multiclass_nms of EVERY image of a batch in one pass: rows of multi_bboxes / multi_scores belong to image
    img_of[row] (int64, rows grouped by image in ascending order).  -> (dets list, labels list), element i bit-identical to
    multiclass_nms(rows of image i): the per-image class shift idx * (max coordinate of that image's candidates + 1), the
    stable score order and the cut to max_num are the per-image ones; (image, class) pairs are the segments of one NMS launch.
    Hard NMS only -- callers route soft_nms (a sequential per-class decay) through the per-image form."""
    from ..mmcv_ops import nms_sorted_mask
    cfg = dict(nms_cfg)
    assert cfg.pop('type', 'nms') == 'nms'
    class_agnostic = cfg.pop('class_agnostic', False)
    cfg.pop('split_thr', None)
    thr = cfg.pop('iou_threshold', cfg.pop('iou_thr', None))
    offset = cfg.pop('offset', 0)
    num_classes = multi_scores.size(1) - 1
    n = multi_scores.size(0)
    dev = multi_scores.device
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(n, -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(n, num_classes, 4)
    scores = multi_scores[:, :-1]
    valid = scores > score_thr
    pos = valid.nonzero(as_tuple=False)                       # row-major: per image, the per-image call's candidate order
    empty = (multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0, ), dtype=torch.long))
    if pos.size(0) == 0:
        return [empty[0]] * num_imgs, [empty[1]] * num_imgs
    boxes, sc, labels = bboxes[valid], scores[valid], pos[:, 1]
    img = img_of[pos[:, 0]]
    if class_agnostic:
        boxes_for_nms, seg_id = boxes, img
    else:
        max_c = torch.full((num_imgs, ), float('-inf'), device=dev, dtype=boxes.dtype)
        max_c.scatter_reduce_(0, img, boxes.max(dim=1)[0], 'amax')
        boxes_for_nms = boxes + (labels.to(boxes) * (max_c[img] + 1))[:, None]
        seg_id = img * num_classes + labels
    order = torch.sort(sc, descending=True, stable=True)[1]
    perm = order[torch.sort(seg_id[order], stable=True)[1]]
    counts = torch.bincount(seg_id, minlength=num_imgs * (1 if class_agnostic else num_classes))
    seg = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=dev)
    seg[1:] = torch.cumsum(counts, 0)
    keep_sorted = nms_sorted_mask(boxes_for_nms[perm], thr, offset, seg, int(counts.max().item()))
    keep_global = torch.zeros(sc.size(0), dtype=torch.bool, device=dev)
    keep_global[perm] = keep_sorted.bool()
    kept = order[keep_global[order]]                          # descending score over the whole batch
    kept = kept[torch.sort(img[kept], stable=True)[1]]        # grouped by image, score order kept inside
    per_img = torch.bincount(img[kept], minlength=num_imgs)
    if max_num > 0:
        start = torch.cumsum(per_img, 0) - per_img
        rank = torch.arange(kept.numel(), device=dev) - start[img[kept]]
        kept = kept[rank < max_num]
        per_img = per_img.clamp(max=max_num)
    dets = torch.cat([boxes[kept], sc[kept, None]], -1)
    sizes = per_img.tolist()
    return list(dets.split(sizes)), list(labels[kept].split(sizes))


def merge_aug_proposals(aug_proposals, img_metas, rpn_test_cfg):
    """merge_augs.py:9-51: proposals (n,5) of every augmentation of ONE image, mapped back to the original image
    scale, de-duplicated by one NMS and cut to max_num.  -> (k,5)."""
    recovered = []
    for proposals, info in zip(aug_proposals, img_metas):
        p = proposals.clone()
        p[:, :4] = bbox_mapping_back(p[:, :4], info['img_shape'], info['scale_factor'], info['flip'],
                                     info['flip_direction'])
        recovered.append(p)
    allp = torch.cat(recovered, dim=0)
    merged, _ = nms(allp[:, :4].contiguous(), allp[:, -1].contiguous(), rpn_test_cfg.nms_thr)
    order = merged[:, 4].sort(0, descending=True)[1]
    return merged[order[:min(rpn_test_cfg.max_num, merged.shape[0])], :]


def merge_aug_bboxes(aug_bboxes, aug_scores, img_metas, rcnn_test_cfg):
    """merge_augs.py:54-81: mean over the augmentations of the boxes (mapped back) and of the scores."""
    recovered = []
    for bboxes, info in zip(aug_bboxes, img_metas):
        m = info[0]
        recovered.append(bbox_mapping_back(bboxes, m['img_shape'], m['scale_factor'], m['flip'], m['flip_direction']))
    bboxes = torch.stack(recovered).mean(dim=0)
    if aug_scores is None:
        return bboxes
    return bboxes, torch.stack(aug_scores).mean(dim=0)


def merge_aug_scores(aug_scores):
    """merge_augs.py:84-89."""
    return torch.mean(torch.stack(aug_scores), dim=0)
