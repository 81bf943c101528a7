#!/usr/bin/env python3
"""GPU kernel time and launch count per phase of the train step and per launching op (torch.profiler, CPU + GPU
activity): where the small kernels come from."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile, record_function

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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with record_function('P:zero_grad'):
        tr.flat.zero_grad()
    with record_function('P:backbone+fpn'):
        x = model.extract_feat(data['img'])
    with record_function('P:rpn_fwd'):
        outs = model.rpn_head(x)
    with record_function('P:rpn_loss'):
        rl = model.rpn_head.loss(*outs, data['gt_bboxes'], data['img_metas'])
    with record_function('P:proposals'):
        props, n_keep = model.rpn_head.get_bboxes(*outs, data['img_metas'], cfg=model.train_cfg.rpn_proposal, padded=True)
    with record_function('P:roi_head'):
        hl = rh.forward_train_static(x, data['img_metas'], props, n_keep, data['gt_bboxes'], data['gt_labels'])
    rl.update(hl)
    with record_function('P:parse'):
        loss, _ = model._parse_losses(rl)
    with record_function('P:backward'):
        loss.backward()
    with record_function('P:optimizer'):
        tr.exchange.finish_step()
        tr.lr_dev.fill_(0.001)
        from htd_amd import mmcv_ops as M
        M.sgd_momentum_step_(tr.flat.flat, tr.flat.grad, tr.flat.momentum, tr.lr_dev, 0.9, 1e-4)
    torch.cuda.synchronize()
ev = prof.events()
phases = [e for e in ev if e.name.startswith('P:')]


def phase_of(t):
    for p in phases:
        if p.time_range.start <= t < p.time_range.end:
            return p.name
    return '?'


def top_op(e):
    """outermost aten:: / autograd-node ancestor below the phase marker"""
    best = e
    cur = e
    while cur.cpu_parent is not None and not cur.cpu_parent.name.startswith('P:'):
        cur = cur.cpu_parent
        if cur.name.startswith('aten::') or 'Backward' in cur.name or 'Function' in cur.name:
            best = cur
    return best.name


per_phase = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
per_op = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if any(c.kernels for c in e.cpu_children):
        continue                                   # count a kernel once, at the innermost launching event
    ph = phase_of(e.time_range.start)
    for k in e.kernels:
        rec = per_phase[ph]
        rec[0] += 1
        rec[1] += k.duration
        if k.duration < 20:
            rec[2] += 1
            rec[3] += k.duration
            op = per_op[(ph, top_op(e), k.name[:60])]
            op[0] += 1
            op[1] += k.duration
print(f'{"phase":16s} {"launches":>8s} {"gpu ms":>8s} | {"<20us":>6s} {"ms":>7s}')
for ph, (n, us, ns, uss) in sorted(per_phase.items(), key=lambda kv: -kv[1][1]):
    print(f'{ph:16s} {n:8d} {us / 1e3:8.2f} | {ns:6d} {uss / 1e3:7.2f}')
print()
for (ph, op, kn), (n, us) in sorted(per_op.items(), key=lambda kv: -kv[1][1])[:90]:
    print(f'{us / 1e3:6.3f} ms {n:4d}x {ph:14s} {op[:44]:44s} {kn}')
