#!/usr/bin/env python3
"""Finer wall-time split of HTDRoIHead.forward_train (device sync after each part)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from htd_amd.configs import build_htd_detector
from htd_amd.runner import synthetic_batch
from htd_amd.core import bbox2roi

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
data = synthetic_batch(4, device=dev)
acc = {}
def lap(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
h = model.roi_head
N = 5
for it in range(N + 2):
    if it == 2:
        acc.clear()
    with torch.no_grad():
        x = model.extract_feat(data['img'])
        outs = model.rpn_head(x)
        props = model.rpn_head.get_bboxes(*outs, data['img_metas'], cfg=model.train_cfg.rpn_proposal)
    x = tuple(t.detach().requires_grad_() for t in x)
    gtb, gtl, metas = data['gt_bboxes'], data['gt_labels'], data['img_metas']
    torch.cuda.synchronize(); t = time.perf_counter()
    sr = h._assign_and_sample(0, props, gtb, gtl, [None] * 4); t = lap('s0 assign+sample', t)
    mc, g = h.glbctx_head(x); lg = h.glbctx_head.loss(mc, gtl); t = lap('sfa', t)
    rois = bbox2roi([r.bboxes for r in sr]); t = lap('s0 bbox2roi', t)
    res = h._bbox_forward(0, x, rois, g); t = lap('s0 extract+head', t)
    tg = h.bbox_head[0].get_targets(sr, gtb, gtl, h.train_cfg[0]); t = lap('s0 targets', t)
    l0 = h.bbox_head[0].loss(res['cls_score'], res['bbox_pred'], rois, *tg); t = lap('s0 loss', t)
    with torch.no_grad():
        rl = torch.where(tg[0] == 80, res['cls_score'][:, :-1].argmax(1), tg[0])
        pl = h.bbox_head[0].refine_bboxes(rois, rl, res['bbox_pred'], [r.pos_is_gt for r in sr], metas)
    t = lap('refine', t)
    sr1 = h._assign_and_sample(1, pl, gtb, gtl, [None] * 4); t = lap('s1 assign+sample', t)
    rois1 = bbox2roi([r.bboxes for r in sr1]); t = lap('s1 bbox2roi', t)
    res1 = h._bbox_forward(1, x, rois1, g, sr1); t = lap('s1 extract+BA+pgraph+head', t)
    tg1 = h.bbox_head[1].get_targets(sr1, gtb, gtl, h.train_cfg[1]); t = lap('s1 targets', t)
    l1 = h.bbox_head[1].loss(res1['cls_score'], res1['bbox_pred'], rois1, *tg1); t = lap('s1 loss', t)
for k, v in acc.items():
    print(f'{k:28s} {v / N * 1e3:8.2f} ms')
print('sum', sum(acc.values()) / N * 1e3)
