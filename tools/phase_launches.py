#!/usr/bin/env python3
"""Kernel launches and host wall time per phase of the train step (torch.profiler, CPU activity only)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
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
rh = model.roi_head
with profile(activities=[ProfilerActivity.CPU]) as prof:
    tr.flat.zero_grad()
    with record_function('P:backbone'):
        x = model.extract_feat(data['img'])
    with record_function('P:rpn_fwd'):
        outs = model.rpn_head(x)
    with record_function('P:rpn_loss'):
        rl = model.rpn_head.loss(*outs, data['gt_bboxes'], data['img_metas'])
    with record_function('P:proposals'):
        props = model.rpn_head.get_bboxes(*outs, data['img_metas'], cfg=model.train_cfg.rpn_proposal)
    with record_function('P:roi_head'):
        hl = rh.forward_train(x, data['img_metas'], props, data['gt_bboxes'], data['gt_labels'])
    rl.update(hl)
    with record_function('P:parse'):
        loss, _ = model._parse_losses(rl)
    with record_function('P:backward'):
        loss.backward()
    torch.cuda.synchronize()
ev = prof.events()
phases = [e for e in ev if e.name.startswith('P:')]
launch = [e for e in ev if 'LaunchKernel' in e.name or e.name in ('hipMemcpyAsync', 'hipMemsetAsync', 'hipMemcpyWithStream')]
names = collections.Counter(e.name for e in ev)
for ph in phases:
    t0, t1 = ph.time_range.start, ph.time_range.end
    n = sum(1 for e in launch if t0 <= e.time_range.start < t1)
    syncs = sum(1 for e in ev if t0 <= e.time_range.start < t1 and ('Synchronize' in e.name or e.name == 'aten::item' or e.name == 'aten::_local_scalar_dense'))
    print(f'{ph.name:14s} host {(t1 - t0) / 1e3:8.2f} ms  launches {n:5d}  syncs/items {syncs}')
if len(sys.argv) > 1:
    ph = [p for p in phases if p.name == 'P:' + sys.argv[1]][0]
    t0, t1 = ph.time_range.start, ph.time_range.end
    c = collections.Counter(e.name for e in ev if t0 <= e.time_range.start < t1 and e.name.startswith('aten::'))
    for k, v in c.most_common(40):
        print(f'   {v:5d} {k}')
