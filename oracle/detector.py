"""oracle/detector.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional CPU (torch fp32) restatement of the HTD detector hot path: ResNet -> FPN ->
RPN (loss + proposals) -> HTDRoIHead (SFA, stage 1, BA, PGraph, stage 2) -> losses /
detections.  Parameters come in as a flat dict with the REFERENCE's state_dict key
names (`backbone.layer1.0.conv1.weight`, `roi_head.bbox_head.1.graph_lvl0_cls.weight`
...), so the same seeded weights drive the reference (when fixtures are generated),
this oracle and the HIP product.

Pinned by tests/golden/{sfa,stage1_head,pgraph,ba,detector}.npz, which were produced
by the reference's own files (tests/golden/make_golden.py).  RoIAlign / NMS / DCN come
from oracle/ops.py and are "parity unpinned" (SURVEY.md section 8c).

Follows the reference loop structure (per image, per level) on purpose.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import boxes as B
from . import ops


def htd_config(depth=50, dcn=False):
    """Hyper-parameters of configs/htd/htd_resnet50_1x.py:5-168 (and the R101 / R101-DCN
    variants configs/htd/htd_resnet101_2x.py, htd_resnet101_dcn_2x_mstrain.py:142)."""
    rcnn = lambda thr: dict(assigner=dict(pos_iou_thr=thr, neg_iou_thr=thr, min_pos_iou=thr,
                                          match_low_quality=False),
                            sampler=dict(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
                            pos_weight=-1)
    return dict(
        depth=depth, dcn=dcn, frozen_stages=1,
        strides=[4, 8, 16, 32, 64], anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0],
        num_classes=80, roi_strides=[4, 8, 16, 32], edge=1, alpha=1,
        stage_loss_weights=[1, 0.5], sfa_loss_weight=3.0,
        stds=[(0.1, 0.1, 0.2, 0.2), (0.05, 0.05, 0.1, 0.1)],
        train_cfg=dict(
            rpn=dict(assigner=dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True),
                     sampler=dict(num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
                     allowed_border=0, pos_weight=-1),
            rpn_proposal=dict(nms_pre=2000, nms_post=2000, max_num=2000, nms_thr=0.7, min_bbox_size=0),
            rcnn=[rcnn(0.5), rcnn(0.6)]),
        test_cfg=dict(rpn=dict(nms_pre=1000, nms_post=1000, max_num=1000, nms_thr=0.7, min_bbox_size=0),
                      rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)))


# ------------------------------------------------------------------ backbone / neck
def _bn_eval(sd, p, x, eps=1e-5):
    """Frozen-statistics BN (norm_eval=True, backbones/resnet.py:640-649)."""
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0., eps)


def _conv2(sd, p, x, stride, dcn, groups=1):
    if dcn and (p + '.conv_offset.weight') in sd:
        off = F.conv2d(x, sd[p + '.conv_offset.weight'], sd[p + '.conv_offset.bias'], stride=stride, padding=1)
        return ops.deform_conv2d_autograd(x, off, sd[p + '.weight'], stride=stride, padding=1, groups=groups)
    return F.conv2d(x, sd[p + '.weight'], None, stride=stride, padding=1, groups=groups)


def bottleneck(sd, p, x, stride, dcn=False, groups=1):
    """Bottleneck.forward, backbones/resnet.py:260-300 (style='pytorch': stride on conv2); groups > 1: the ResNeXt
    block of backbones/resnext.py:9-84 (same forward, grouped conv2)."""
    out = F.relu(_bn_eval(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'])))
    out = F.relu(_bn_eval(sd, p + '.bn2', _conv2(sd, p + '.conv2', out, stride, dcn, groups)))
    out = _bn_eval(sd, p + '.bn3', F.conv2d(out, sd[p + '.conv3.weight']))
    idt = x
    if (p + '.downsample.0.weight') in sd:
        idt = _bn_eval(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride))
    return F.relu(out + idt)


def resnet(sd, x, depth=50, dcn=False, prefix='backbone.'):
    """ResNet.forward, backbones/resnet.py:623-638."""
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
    x = F.relu(_bn_eval(sd, prefix + 'bn1', F.conv2d(x, sd[prefix + 'conv1.weight'], stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for i, nb in enumerate(blocks):
        for j in range(nb):
            x = bottleneck(sd, f'{prefix}layer{i + 1}.{j}', x, 2 if (j == 0 and i > 0) else 1, dcn and i > 0)
        outs.append(x)
    return outs


def fpn(sd, feats, prefix='neck.'):
    """FPN.forward, necks/fpn.py:165-216 (num_outs=5, no extra convs)."""
    lats = [F.conv2d(f, sd[f'{prefix}lateral_convs.{i}.conv.weight'], sd[f'{prefix}lateral_convs.{i}.conv.bias'])
            for i, f in enumerate(feats)]
    for i in range(len(lats) - 1, 0, -1):
        lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode='nearest')
    outs = [F.conv2d(l, sd[f'{prefix}fpn_convs.{i}.conv.weight'], sd[f'{prefix}fpn_convs.{i}.conv.bias'], padding=1)
            for i, l in enumerate(lats)]
    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
    return outs


# ------------------------------------------------------------------ RPN
def rpn_forward(sd, feats, prefix='rpn_head.'):
    """RPNHead.forward_single, dense_heads/rpn_head.py:37-43."""
    cls, reg = [], []
    for x in feats:
        x = F.relu(F.conv2d(x, sd[prefix + 'rpn_conv.weight'], sd[prefix + 'rpn_conv.bias'], padding=1))
        cls.append(F.conv2d(x, sd[prefix + 'rpn_cls.weight'], sd[prefix + 'rpn_cls.bias']))
        reg.append(F.conv2d(x, sd[prefix + 'rpn_reg.weight'], sd[prefix + 'rpn_reg.bias']))
    return cls, reg


def rpn_targets_single(flat_anchors, valid, gt_bboxes, img_meta, cfg):
    """AnchorHead._get_targets_single, dense_heads/anchor_head.py:172-269 (RPN: gt_labels None)."""
    inside = B.anchor_inside_flags(flat_anchors, valid, img_meta['img_shape'][:2], cfg['allowed_border'])
    anchors = flat_anchors[inside, :]
    ar = B.max_iou_assign(anchors, gt_bboxes, None, **cfg['assigner'])
    sr = B.random_sample(ar, anchors, gt_bboxes, None, **cfg['sampler'])
    n = anchors.shape[0]
    bbox_targets, bbox_weights = torch.zeros_like(anchors), torch.zeros_like(anchors)
    labels = anchors.new_full((n, ), 1, dtype=torch.long)  # num_classes == 1 -> bg label 1
    label_weights = anchors.new_zeros(n)
    if len(sr.pos_inds) > 0:
        bbox_targets[sr.pos_inds, :] = B.bbox2delta(sr.pos_bboxes, sr.pos_gt_bboxes)
        bbox_weights[sr.pos_inds, :] = 1.0
        labels[sr.pos_inds] = 0
        label_weights[sr.pos_inds] = 1.0 if cfg['pos_weight'] <= 0 else cfg['pos_weight']
    if len(sr.neg_inds) > 0:
        label_weights[sr.neg_inds] = 1.0
    total = flat_anchors.size(0)

    def unmap(data, fill=0):
        ret = data.new_full((total, ) + tuple(data.shape[1:]), fill)
        ret[inside] = data
        return ret
    return unmap(labels, 1), unmap(label_weights), unmap(bbox_targets), unmap(bbox_weights), sr.pos_inds, sr.neg_inds


def rpn_loss(cls_scores, bbox_preds, gt_bboxes, img_metas, cfg, strides):
    """AnchorHead.loss / loss_single, anchor_head.py:373-488 via RPNHead.loss rpn_head.py:45-76."""
    sizes = [c.shape[-2:] for c in cls_scores]
    anchors = B.grid_anchors(sizes, strides, cfg['anchor_scales'], cfg['anchor_ratios'])
    num_lvl = [a.size(0) for a in anchors]
    flat = torch.cat(anchors)
    per_img = []
    for i, meta in enumerate(img_metas):
        valid = torch.cat(B.valid_flags(sizes, strides, meta['pad_shape']))
        per_img.append(rpn_targets_single(flat, valid, gt_bboxes[i], meta, cfg['train_cfg']['rpn']))
    num_total = sum(max(t[4].numel(), 1) for t in per_img) + sum(max(t[5].numel(), 1) for t in per_img)

    def to_levels(k):
        stacked = torch.stack([t[k] for t in per_img], 0)
        out, s = [], 0
        for n in num_lvl:
            out.append(stacked[:, s:s + n])
            s += n
        return out
    labels, lweights, btargets, bweights = (to_levels(k) for k in range(4))
    losses_cls, losses_bbox = [], []
    for l in range(len(cls_scores)):
        cs = cls_scores[l].permute(0, 2, 3, 1).reshape(-1, 1)
        losses_cls.append(B.binary_cross_entropy(cs, labels[l].reshape(-1), lweights[l].reshape(-1),
                                                 avg_factor=num_total))
        bp = bbox_preds[l].permute(0, 2, 3, 1).reshape(-1, 4)
        losses_bbox.append(B.smooth_l1_loss(bp, btargets[l].reshape(-1, 4), bweights[l].reshape(-1, 4),
                                            beta=1.0 / 9.0, avg_factor=num_total))
    return dict(loss_rpn_cls=losses_cls, loss_rpn_bbox=losses_bbox)


def rpn_get_bboxes(cls_scores, bbox_preds, img_metas, pcfg, cfg, strides, trace=None):
    """AnchorHead.get_bboxes anchor_head.py:491-579 + RPNHead._get_bboxes_single rpn_head.py:78-168.
    trace (list): per image (keep rows of the level-concatenated candidate list, flat anchor index of each kept box,
    then what the NMS was fed: decoded candidate boxes, scores, level ids, flat anchor index of every candidate)."""
    sizes = [c.shape[-2:] for c in cls_scores]
    level_off = [0]
    for c in cls_scores:
        level_off.append(level_off[-1] + int(c.shape[1] * c.shape[2] * c.shape[3]))
    mlvl_anchors = B.grid_anchors(sizes, strides, cfg['anchor_scales'], cfg['anchor_ratios'])
    results = []
    for img_id, meta in enumerate(img_metas):
        scores_l, preds_l, anch_l, ids_l, flat_l = [], [], [], [], []
        for idx in range(len(cls_scores)):
            s = cls_scores[idx][img_id].detach().permute(1, 2, 0).reshape(-1).sigmoid()
            p = bbox_preds[idx][img_id].detach().permute(1, 2, 0).reshape(-1, 4)
            a = mlvl_anchors[idx]
            topk = torch.arange(s.shape[0])
            if pcfg['nms_pre'] > 0 and s.shape[0] > pcfg['nms_pre']:
                ranked, rank_inds = s.sort(descending=True, stable=True)
                topk = rank_inds[:pcfg['nms_pre']]
                s, p, a = ranked[:pcfg['nms_pre']], p[topk, :], a[topk, :]
            flat_l.append(topk + level_off[idx])
            scores_l.append(s)
            preds_l.append(p)
            anch_l.append(a)
            ids_l.append(s.new_full((s.size(0), ), idx, dtype=torch.long))
        scores, anchors, preds, ids = torch.cat(scores_l), torch.cat(anch_l), torch.cat(preds_l), torch.cat(ids_l)
        proposals = B.delta2bbox(anchors, preds, max_shape=meta['img_shape'])
        if pcfg['min_bbox_size'] > 0:
            w, h = proposals[:, 2] - proposals[:, 0], proposals[:, 3] - proposals[:, 1]
            v = (w >= pcfg['min_bbox_size']) & (h >= pcfg['min_bbox_size'])
            proposals, scores, ids = proposals[v], scores[v], ids[v]
            flat_l = [torch.cat(flat_l)[v]]
        dets, keep = ops.batched_nms(proposals, scores, ids, dict(type='nms', iou_threshold=pcfg['nms_thr']))
        if trace is not None:
            trace.append((keep[:pcfg['nms_post']], torch.cat(flat_l)[keep[:pcfg['nms_post']]], proposals, scores, ids,
                          torch.cat(flat_l)))
        results.append(dets[:pcfg['nms_post']])
    return results


# ------------------------------------------------------------------ RoI extractors
def single_roi_extract(feats, rois, strides=(4, 8, 16, 32)):
    """SingleRoIExtractor.forward, roi_extractors/single_level_roi_extractor.py:53-99."""
    out = feats[0].new_zeros(rois.size(0), feats[0].size(1), 7, 7)
    lvls = B.map_roi_levels(rois, len(strides))
    for i, s in enumerate(strides):
        inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel() > 0:
            out = out.index_put((inds, ), ops.roi_align(feats[i], rois[inds], 7, 1.0 / s, 0, True))
    return out


def ba_extract(sd, feats, rois, strides=(4, 8, 16, 32), edge=1, prefix='roi_head.bbox_roi_extractor.1.'):
    """AdptRoIExtractor.forward (BA), roi_extractors/adaptative_roi_extractor.py:49-91.
    The reference's .squeeze() breaks n == 1 (:73); here n == 1 keeps its axis (SURVEY B, 'F')."""
    n = rois.size(0)
    if n == 0:
        return feats[0].new_zeros(0, feats[0].size(1), 7, 7)
    roi_feat, atts = [], []
    for i, s in enumerate(strides):
        f = ops.roi_align(feats[i], rois, 7, 1.0 / s, 0, True)
        a = F.adaptive_avg_pool2d(f, 1)
        a = torch.tanh(F.conv2d(a, sd[prefix + 'conv1.weight'], sd[prefix + 'conv1.bias']))
        a = F.conv2d(a, sd[prefix + 'conv2.weight'], sd[prefix + 'conv2.bias'])
        atts.append(a.reshape(1, n))
        roi_feat.append(f.unsqueeze(0))
    roi_feat = torch.cat(roi_feat, 0)
    atts = torch.cat(atts, 0).softmax(0)
    fused = (atts.view(len(strides), n, 1, 1, 1) * roi_feat).sum(0)
    border = ops.roi_align(feats[0], rois, 7, 1.0 / strides[0], 0, True)
    mask = torch.ones(7, 7)
    mask[edge:-edge, edge:-edge] = 0
    return fused + border * mask


# ------------------------------------------------------------------ heads
def sfa_forward(sd, feats, prefix='roi_head.glbctx_head.'):
    """GlobalContextHead.forward, bbox_heads/global_context_head.py:382-392."""
    x = feats[-1]
    for i in range(4):
        x = F.relu(F.conv2d(x, sd[f'{prefix}convs.{i}.conv.weight'], sd[f'{prefix}convs.{i}.conv.bias'], padding=1))
    x = F.adaptive_avg_pool2d(x, 1)
    mc = F.linear(x.reshape(x.size(0), -1), sd[prefix + 'fc.weight'], sd[prefix + 'fc.bias'])
    return mc, x


def sfa_loss(pred, labels, loss_weight=3.0):
    """GlobalContextHead.loss, global_context_head.py:394-401."""
    targets = pred.new_zeros(pred.size())
    for i, l in enumerate(labels):
        targets[i, l.unique()] = 1.0
    return loss_weight * F.binary_cross_entropy_with_logits(pred, targets)


def fuse_global(roi_feats, global_feat, rois):
    """HTDRoIHead._fuse_global htd_roi_head.py:133-141 (== htd_bbox_head.py:147-155)."""
    return roi_feats + global_feat[rois[:, 0].long()]


def shared2fc_forward(sd, x, prefix='roi_head.bbox_head.0.'):
    """Shared2FCBBoxHead.forward, bbox_heads/convfc_bbox_head.py:135-173."""
    x = x.flatten(1)
    x = F.relu(F.linear(x, sd[prefix + 'shared_fcs.0.weight'], sd[prefix + 'shared_fcs.0.bias']))
    x = F.relu(F.linear(x, sd[prefix + 'shared_fcs.1.weight'], sd[prefix + 'shared_fcs.1.bias']))
    return (F.linear(x, sd[prefix + 'fc_cls.weight'], sd[prefix + 'fc_cls.bias']),
            F.linear(x, sd[prefix + 'fc_reg.weight'], sd[prefix + 'fc_reg.bias']))


def htd_bbox_head_forward(sd, x_cls, x_reg, rois, enhanced, pos_rois, global_feat, alpha=1,
                          prefix='roi_head.bbox_head.1.', prefix0='roi_head.bbox_head.0.', num_levels=4):
    """HTDBBoxHead.forward (PGraph + BA-consuming reg branch), bbox_heads/htd_bbox_head.py:157-230
    with relpace=False, average=False (configs/htd/htd_resnet50_1x.py:75-80)."""
    w0, b0 = sd[prefix0 + 'fc_cls.weight'], sd[prefix0 + 'fc_cls.bias']
    prototype = torch.cat((w0, b0.unsqueeze(1)), 1).detach()
    bs = int(torch.max(rois[:, 0])) + 1

    def fcs(t):
        t = F.relu(F.linear(t, sd[prefix + 'fcs.0.weight'], sd[prefix + 'fcs.0.bias']))
        return F.relu(F.linear(t, sd[prefix + 'fcs.2.weight'], sd[prefix + 'fcs.2.bias']))
    x_cls_glb = fcs(fuse_global(x_cls, global_feat, rois).flatten(1))
    x_reg = fuse_global(x_reg, global_feat, pos_rois)
    x_reg = x_reg + alpha * enhanced
    # reg branch: 4 ConvModules (3x3 no bias; GN36+ReLU x3, ReLU-only last) :77-113,186
    for i in range(4):
        x_reg = F.conv2d(x_reg, sd[f'{prefix}convs.{i}.conv.weight'], None, padding=1)
        if i < 3:
            x_reg = F.group_norm(x_reg, 36, sd[f'{prefix}convs.{i}.gn.weight'], sd[f'{prefix}convs.{i}.gn.bias'], 1e-5)
        x_reg = F.relu(x_reg)
    x_reg = F.avg_pool2d(x_reg, 7).view(x_reg.size(0), -1)
    # cls branch
    x = fcs(x_cls.flatten(1))
    sam = torch.mm(F.linear(x, w0, b0).softmax(-1), prototype)
    lvls = B.map_roi_levels(rois, num_levels)
    refined = x.new_zeros(x.size(0), 1024)
    for b in range(bs):
        for i in range(num_levels):
            idx = ((lvls == i) & (rois[:, 0] == b)).nonzero(as_tuple=False).squeeze(1)
            if idx.numel() == 0:
                continue
            sam_, rois_ = sam[idx], rois[idx, 1:5]
            M = B.bbox_overlaps(rois_, rois_).fill_diagonal_(1.)
            M = (M > 0).to(x.dtype)
            D = torch.diag(M.sum(-1).pow(-0.5))
            A_local = torch.mm(torch.mm(D, M), D)
            mixed = torch.mm(A_local, x[idx])
            A_global = ((1. - M) * torch.mm(sam_, sam_.t())).softmax(-1)
            new = F.relu(F.linear(torch.matmul(A_global, mixed), sd[f'{prefix}graph_lvl{i}_cls.weight'],
                                  sd[f'{prefix}graph_lvl{i}_cls.bias']))
            refined = refined.index_put((idx, ), new)
    feat_new = x_cls_glb + refined
    return (F.linear(feat_new, sd[prefix + 'fc_cls.weight'], sd[prefix + 'fc_cls.bias']),
            F.linear(x_reg, sd[prefix + 'fc_reg.weight'], sd[prefix + 'fc_reg.bias']))


def bbox_targets(sampling_results, stds, num_classes=80, pos_weight=-1):
    """BBoxHead.get_targets/_get_target_single, bbox_heads/bbox_head.py:85-139."""
    L, LW, BT, BW = [], [], [], []
    for r in sampling_results:
        npos, nneg = r.pos_bboxes.size(0), r.neg_bboxes.size(0)
        n = npos + nneg
        labels = r.pos_bboxes.new_full((n, ), num_classes, dtype=torch.long)
        lw, bt, bw = r.pos_bboxes.new_zeros(n), r.pos_bboxes.new_zeros(n, 4), r.pos_bboxes.new_zeros(n, 4)
        if npos > 0:
            labels[:npos] = r.pos_gt_labels
            lw[:npos] = 1.0 if pos_weight <= 0 else pos_weight
            bt[:npos] = B.bbox2delta(r.pos_bboxes, r.pos_gt_bboxes, (0., 0., 0., 0.), stds)
            bw[:npos] = 1
        if nneg > 0:
            lw[-nneg:] = 1.0
        L.append(labels), LW.append(lw), BT.append(bt), BW.append(bw)
    return torch.cat(L), torch.cat(LW), torch.cat(BT), torch.cat(BW)


def bbox_loss(cls_score, bbox_pred, labels, label_weights, bbox_tgts, bbox_weights, num_classes=80):
    """BBoxHead.loss, bbox_head.py:142-186 (reg_class_agnostic=True)."""
    losses = dict()
    avg = max(torch.sum(label_weights > 0).float().item(), 1.)
    losses['loss_cls'] = B.cross_entropy(cls_score, labels, label_weights, avg_factor=avg)
    losses['acc'] = B.accuracy(cls_score, labels)
    pos = (labels >= 0) & (labels < num_classes)
    if pos.any():
        losses['loss_bbox'] = B.smooth_l1_loss(bbox_pred.view(-1, 4)[pos], bbox_tgts[pos], bbox_weights[pos],
                                               beta=1.0, avg_factor=bbox_tgts.size(0))
    else:
        losses['loss_bbox'] = bbox_pred[pos].sum()
    return losses


def regress_by_class(rois, bbox_pred, img_meta, stds):
    """BBoxHead.regress_by_class, bbox_head.py:306-335 (class agnostic)."""
    if rois.size(1) == 4:
        return B.delta2bbox(rois, bbox_pred, (0., 0., 0., 0.), stds, max_shape=img_meta['img_shape'])
    bb = B.delta2bbox(rois[:, 1:], bbox_pred, (0., 0., 0., 0.), stds, max_shape=img_meta['img_shape'])
    return torch.cat((rois[:, [0]], bb), dim=1)


def refine_bboxes(rois, bbox_preds, pos_is_gts, img_metas, stds):
    """BBoxHead.refine_bboxes, bbox_head.py:227-304."""
    out = []
    for i in range(len(img_metas)):
        inds = torch.nonzero(rois[:, 0] == i, as_tuple=False).squeeze(1)
        bb = regress_by_class(rois[inds, 1:], bbox_preds[inds], img_metas[i], stds)
        keep = pos_is_gts[i].new_ones(inds.numel())
        keep[:len(pos_is_gts[i])] = 1 - pos_is_gts[i]
        out.append(bb[keep.type(torch.bool)])
    return out


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1):
    """mmdet/core/post_processing/bbox_nms.py:7-71 (class-agnostic boxes (n,4))."""
    num_classes = multi_scores.size(1) - 1
    bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    valid = scores > score_thr
    bboxes = bboxes[valid]
    scores = scores[valid]
    labels = valid.nonzero(as_tuple=False)[:, 1]
    if bboxes.numel() == 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0, ), dtype=torch.long)
    dets, keep = ops.batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep]


# ------------------------------------------------------------------ HTDRoIHead
def _assign_sample(proposals, gt_bboxes, gt_labels, rcfg):
    out = []
    for j in range(len(proposals)):
        ar = B.max_iou_assign(proposals[j], gt_bboxes[j], gt_labels[j], **rcfg['assigner'])
        out.append(B.random_sample(ar, proposals[j], gt_bboxes[j], gt_labels[j], **rcfg['sampler']))
    return out


def roi_head_forward_train(sd, x, img_metas, proposal_list, gt_bboxes, gt_labels, cfg, trace=None):
    """HTDRoIHead.forward_train, roi_heads/htd_roi_head.py:217-317 (stage-2 positives generalised
    from image ids {0,1} to every image, identical for B <= 2: SURVEY.md fact 3)."""
    losses = dict()
    nc = cfg['num_classes']
    tc = cfg['train_cfg']['rcnn']
    sr0 = _assign_sample(proposal_list, gt_bboxes, gt_labels, tc[0])
    mc_pred, gfeat = sfa_forward(sd, x)
    losses['loss_global'] = sfa_loss(mc_pred, gt_labels, cfg['sfa_loss_weight'])
    # stage 1
    rois = B.bbox2roi([r.bboxes for r in sr0])
    feats = fuse_global(single_roi_extract(x[:4], rois, cfg['roi_strides']), gfeat, rois)
    cls0, reg0 = shared2fc_forward(sd, feats)
    tg0 = bbox_targets(sr0, cfg['stds'][0], nc)
    for k, v in bbox_loss(cls0, reg0, *tg0, num_classes=nc).items():
        losses['s0.' + k] = v * cfg['stage_loss_weights'][0] if 'loss' in k else v
    with torch.no_grad():
        props = refine_bboxes(rois, reg0, [r.pos_is_gt for r in sr0], img_metas, cfg['stds'][0])
    # stage 2
    sr1 = _assign_sample(props, gt_bboxes, gt_labels, tc[1])
    rois = B.bbox2roi([r.bboxes for r in sr1])
    pos_rois = B.bbox2roi([r.pos_bboxes for r in sr1])
    bbox_feats = single_roi_extract(x[:4], rois, cfg['roi_strides'])
    enhanced = ba_extract(sd, x[:4], pos_rois, cfg['roi_strides'], cfg['edge'])
    pos_rows, start = [], 0
    for r in sr1:
        pos_rows.append(torch.arange(start, start + r.pos_bboxes.size(0)))
        start += r.pos_bboxes.size(0) + r.neg_bboxes.size(0)
    pos_rows = torch.cat(pos_rows)
    cls1, reg1p = htd_bbox_head_forward(sd, bbox_feats, bbox_feats[pos_rows], rois, enhanced, pos_rois, gfeat,
                                        cfg['alpha'])
    reg1 = cls1.new_zeros(cls1.size(0), 4).index_put((pos_rows, ), reg1p)
    tg1 = bbox_targets(sr1, cfg['stds'][1], nc)
    for k, v in bbox_loss(cls1, reg1, *tg1, num_classes=nc).items():
        losses['s1.' + k] = v * cfg['stage_loss_weights'][1] if 'loss' in k else v
    if trace is not None:
        trace.update(rois0=B.bbox2roi([r.bboxes for r in sr0]), cls0=cls0, reg0=reg0, rois1=rois, cls1=cls1,
                     reg1=reg1, pos_rows=pos_rows, global_feat=gfeat, mc_pred=mc_pred,
                     samples=[[(r.pos_inds, r.neg_inds) for r in sr] for sr in (sr0, sr1)])
    return losses


def roi_head_simple_test(sd, x, proposal_list, img_metas, cfg, trace=None):
    """HTDRoIHead.simple_test, htd_roi_head.py:319-386 -> list (per image) of (dets (k,5), labels (k,))."""
    n_per = [len(p) for p in proposal_list]
    rois = B.bbox2roi(proposal_list)
    _, gfeat = sfa_forward(sd, x)
    feats = fuse_global(single_roi_extract(x[:4], rois, cfg['roi_strides']), gfeat, rois)
    cls0, reg0 = shared2fc_forward(sd, feats)
    rois = torch.cat([regress_by_class(r, p, m, cfg['stds'][0])
                      for r, p, m in zip(rois.split(n_per), reg0.split(n_per), img_metas)])
    bbox_feats = single_roi_extract(x[:4], rois, cfg['roi_strides'])
    enhanced = ba_extract(sd, x[:4], rois, cfg['roi_strides'], cfg['edge'])
    cls1, reg1 = htd_bbox_head_forward(sd, bbox_feats, bbox_feats, rois, enhanced, rois, gfeat, cfg['alpha'])
    if trace is not None:
        trace.update(rois0=B.bbox2roi(proposal_list), cls0=cls0, reg0=reg0, rois1=rois, cls1=cls1, reg1=reg1)
    out = []
    rc = cfg['test_cfg']['rcnn']
    for r, c0, c1, p, m in zip(rois.split(n_per), cls0.split(n_per), cls1.split(n_per), reg1.split(n_per), img_metas):
        scores = F.softmax((c0 + c1) / 2.0, dim=1)
        bboxes = B.delta2bbox(r[:, 1:], p, (0., 0., 0., 0.), cfg['stds'][1], max_shape=m['img_shape'])
        out.append(multiclass_nms(bboxes, scores, rc['score_thr'], rc['nms'], rc['max_per_img']))
    return out


# ------------------------------------------------------------------ detector
def extract_feat(sd, img, cfg):
    return fpn(sd, resnet(sd, img, cfg['depth'], cfg['dcn']))


def forward_train(sd, img, img_metas, gt_bboxes, gt_labels, cfg, trace=None):
    """TwoStageDetector.forward_train, detectors/two_stage.py:107-170."""
    x = extract_feat(sd, img, cfg)
    cls, reg = rpn_forward(sd, x)
    losses = rpn_loss(cls, reg, gt_bboxes, img_metas, cfg, cfg['strides'])
    proposals = rpn_get_bboxes(cls, reg, img_metas, cfg['train_cfg']['rpn_proposal'], cfg, cfg['strides'])
    if trace is not None:
        trace.update(feats=x, rpn_cls=cls, rpn_reg=reg, proposals=proposals)
    losses.update(roi_head_forward_train(sd, x, img_metas, proposals, gt_bboxes, gt_labels, cfg, trace))
    return losses


def parse_losses(losses):
    """BaseDetector._parse_losses, detectors/base.py:184-223 (single process)."""
    log_vars = {}
    for k, v in losses.items():
        log_vars[k] = v.mean() if isinstance(v, torch.Tensor) else sum(t.mean() for t in v)
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, {k: float(v.detach()) for k, v in log_vars.items()}


def simple_test(sd, img, img_metas, cfg):
    """TwoStageDetector.simple_test, two_stage.py:190-211 -> (proposals, [(dets, labels)])."""
    x = extract_feat(sd, img, cfg)
    cls, reg = rpn_forward(sd, x)
    props = rpn_get_bboxes(cls, reg, img_metas, cfg['test_cfg']['rpn'], cfg, cfg['strides'])
    return props, roi_head_simple_test(sd, x, props, img_metas, cfg)


# ------------------------------------------------------------------ test-time augmentation
def bbox_flip(bboxes, img_shape, direction='horizontal'):
    """core/bbox/transforms.py:6-31."""
    out = bboxes.clone()
    if direction in ('horizontal', 'diagonal'):
        out[..., 0::4] = img_shape[1] - bboxes[..., 2::4]
        out[..., 2::4] = img_shape[1] - bboxes[..., 0::4]
    if direction in ('vertical', 'diagonal'):
        out[..., 1::4] = img_shape[0] - bboxes[..., 3::4]
        out[..., 3::4] = img_shape[0] - bboxes[..., 1::4]
    return out


def bbox_mapping(bboxes, meta):
    """core/bbox/transforms.py:34-43."""
    out = bboxes * bboxes.new_tensor(meta['scale_factor'])
    return bbox_flip(out, meta['img_shape'], meta['flip_direction']) if meta['flip'] else out


def bbox_mapping_back(bboxes, meta):
    """core/bbox/transforms.py:46-55."""
    out = bbox_flip(bboxes, meta['img_shape'], meta['flip_direction']) if meta['flip'] else bboxes
    return (out.view(-1, 4) / out.new_tensor(meta['scale_factor'])).view(bboxes.shape)


def merge_aug_proposals(aug_proposals, metas, rcfg):
    """core/post_processing/merge_augs.py:9-51."""
    rec = []
    for p, m in zip(aug_proposals, metas):
        q = p.clone()
        q[:, :4] = bbox_mapping_back(q[:, :4], m)
        rec.append(q)
    allp = torch.cat(rec, 0)
    merged, _ = ops.nms(allp[:, :4].contiguous(), allp[:, -1].contiguous(), rcfg['nms_thr'])
    order = merged[:, 4].sort(0, descending=True)[1]
    return merged[order[:min(rcfg['max_num'], merged.shape[0])]]


def aug_test(sd, imgs, img_metas, cfg):
    """TwoStageDetector.aug_test (two_stage.py:213-222) = extract_feats + RPNTestMixin.aug_test_rpn
    (rpn_test_mixin.py:39-59) + HTDRoIHead.aug_test (htd_roi_head.py:388-433), one image, bbox branch.
    imgs: list (augmentations) of (1,3,H,W); img_metas: list of [meta].  -> (merged proposals, (dets, labels))."""
    feats = [extract_feat(sd, im, cfg) for im in imgs]
    aug_props = []
    for x, metas in zip(feats, img_metas):
        cls, reg = rpn_forward(sd, x)
        aug_props.append(rpn_get_bboxes(cls, reg, metas, cfg['test_cfg']['rpn'], cfg, cfg['strides'])[0])
    proposals = merge_aug_proposals(aug_props, [m[0] for m in img_metas], cfg['test_cfg']['rpn'])
    aug_b, aug_s = [], []
    for x, metas in zip(feats, img_metas):
        m = metas[0]
        rois = B.bbox2roi([bbox_mapping(proposals[:, :4], m)])
        _, gfeat = sfa_forward(sd, x)
        f0 = fuse_global(single_roi_extract(x[:4], rois, cfg['roi_strides']), gfeat, rois)
        cls0, reg0 = shared2fc_forward(sd, f0)
        rois = regress_by_class(rois, reg0, m, cfg['stds'][0])
        bf = single_roi_extract(x[:4], rois, cfg['roi_strides'])
        enh = ba_extract(sd, x[:4], rois, cfg['roi_strides'], cfg['edge'])
        cls1, reg1 = htd_bbox_head_forward(sd, bf, bf, rois, enh, rois, gfeat, cfg['alpha'])
        aug_s.append(F.softmax((cls0 + cls1) / 2.0, dim=1))
        aug_b.append(bbox_mapping_back(B.delta2bbox(rois[:, 1:], reg1, (0., 0., 0., 0.), cfg['stds'][1],
                                                    max_shape=m['img_shape']), m))
    rc = cfg['test_cfg']['rcnn']
    bboxes, scores = torch.stack(aug_b).mean(0), torch.stack(aug_s).mean(0)
    return proposals, multiclass_nms(bboxes, scores, rc['score_thr'], rc['nms'], rc['max_per_img'])


def state_shapes(depth=50, dcn=False, num_classes=80):
    """Key -> shape of every parameter/buffer, named as the reference's state_dict."""
    s = {}

    def bn(p, c):
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            s[f'{p}.{k}'] = (c, )
    s['backbone.conv1.weight'] = (64, 3, 7, 7)
    bn('backbone.bn1', 64)
    inpl = 64
    for i, nb in enumerate({50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]):
        pl = 64 * 2 ** i
        for j in range(nb):
            p = f'backbone.layer{i + 1}.{j}'
            s[p + '.conv1.weight'] = (pl, inpl, 1, 1)
            bn(p + '.bn1', pl)
            s[p + '.conv2.weight'] = (pl, pl, 3, 3)
            if dcn and i > 0:
                s[p + '.conv2.conv_offset.weight'] = (18, pl, 3, 3)
                s[p + '.conv2.conv_offset.bias'] = (18, )
            bn(p + '.bn2', pl)
            s[p + '.conv3.weight'] = (pl * 4, pl, 1, 1)
            bn(p + '.bn3', pl * 4)
            if j == 0:
                s[p + '.downsample.0.weight'] = (pl * 4, inpl, 1, 1)
                bn(p + '.downsample.1', pl * 4)
            inpl = pl * 4
    for i, c in enumerate((256, 512, 1024, 2048)):
        s[f'neck.lateral_convs.{i}.conv.weight'] = (256, c, 1, 1)
        s[f'neck.lateral_convs.{i}.conv.bias'] = (256, )
        s[f'neck.fpn_convs.{i}.conv.weight'] = (256, 256, 3, 3)
        s[f'neck.fpn_convs.{i}.conv.bias'] = (256, )
    s['rpn_head.rpn_conv.weight'], s['rpn_head.rpn_conv.bias'] = (256, 256, 3, 3), (256, )
    s['rpn_head.rpn_cls.weight'], s['rpn_head.rpn_cls.bias'] = (3, 256, 1, 1), (3, )
    s['rpn_head.rpn_reg.weight'], s['rpn_head.rpn_reg.bias'] = (12, 256, 1, 1), (12, )
    e = 'roi_head.bbox_roi_extractor.1.'
    s[e + 'conv1.weight'], s[e + 'conv1.bias'] = (128, 256, 1, 1), (128, )
    s[e + 'conv2.weight'], s[e + 'conv2.bias'] = (1, 128, 1, 1), (1, )
    h = 'roi_head.bbox_head.0.'
    s[h + 'fc_cls.weight'], s[h + 'fc_cls.bias'] = (num_classes + 1, 1024), (num_classes + 1, )
    s[h + 'fc_reg.weight'], s[h + 'fc_reg.bias'] = (4, 1024), (4, )
    s[h + 'shared_fcs.0.weight'], s[h + 'shared_fcs.0.bias'] = (1024, 12544), (1024, )
    s[h + 'shared_fcs.1.weight'], s[h + 'shared_fcs.1.bias'] = (1024, 1024), (1024, )
    h = 'roi_head.bbox_head.1.'
    s[h + 'fc_cls.weight'], s[h + 'fc_cls.bias'] = (num_classes + 1, 1024), (num_classes + 1, )
    s[h + 'fc_reg.weight'], s[h + 'fc_reg.bias'] = (4, 1024), (4, )
    for i, (ci, co) in enumerate(((256, 576), (576, 576), (576, 576), (576, 1024))):
        s[f'{h}convs.{i}.conv.weight'] = (co, ci, 3, 3)
        if i < 3:
            s[f'{h}convs.{i}.gn.weight'], s[f'{h}convs.{i}.gn.bias'] = (co, ), (co, )
    s[h + 'fcs.0.weight'], s[h + 'fcs.0.bias'] = (1024, 12544), (1024, )
    s[h + 'fcs.2.weight'], s[h + 'fcs.2.bias'] = (1024, 1024), (1024, )
    for i in range(4):
        s[f'{h}graph_lvl{i}_cls.weight'], s[f'{h}graph_lvl{i}_cls.bias'] = (1024, 1024), (1024, )
    g = 'roi_head.glbctx_head.'
    for i in range(4):
        s[f'{g}convs.{i}.conv.weight'], s[f'{g}convs.{i}.conv.bias'] = (256, 256, 3, 3), (256, )
    s[g + 'fc.weight'], s[g + 'fc.bias'] = (num_classes + 1, 256), (num_classes + 1, )
    return s
