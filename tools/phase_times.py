#!/usr/bin/env python3
"""Wall time of the phases of one train step (with a device sync after each phase)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
acc = {}
def lap(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
N = 5
for _ in range(N):
    tr.flat.zero_grad()
    t = time.perf_counter()
    x = model.extract_feat(data['img']); t = lap('backbone+fpn fwd', t)
    outs = model.rpn_head(x); t = lap('rpn fwd', t)
    rl = model.rpn_head.loss(*outs, data['gt_bboxes'], data['img_metas']); t = lap('rpn loss', t)
    props = model.rpn_head.get_bboxes(*outs, data['img_metas'], cfg=model.train_cfg.rpn_proposal); t = lap('proposals', t)
    hl = model.roi_head.forward_train(x, data['img_metas'], props, data['gt_bboxes'], data['gt_labels']); t = lap('roi head fwd', t)
    rl.update(hl)
    loss, _ = model._parse_losses(rl); t = lap('parse', t)
    loss.backward(); t = lap('backward', t)
for k, v in acc.items():
    print(f'{k:22s} {v / N * 1e3:8.2f} ms')
print('sum', sum(acc.values()) / N * 1e3)
