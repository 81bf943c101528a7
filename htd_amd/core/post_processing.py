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
