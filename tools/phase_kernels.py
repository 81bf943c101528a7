#!/usr/bin/env python3
"""Device kernels (count, total device time) per logical phase of the train step (torch.profiler, CUDA activity)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch
import htd_amd.core.bbox as cb
import htd_amd.detector.pgraph as pg
import htd_amd.detector.htd_roi_head as rh_mod

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)


def wrap(owner, name, tag=None):
    fn = getattr(owner, name)
    tag = tag or name

    def inner(*a, **k):
        with record_function('Q:' + tag):
            return fn(*a, **k)
    setattr(owner, name, inner)


wrap(rh_mod.HTDRoIHead, '_static_targets')
import htd_amd.core.bbox
wrap(htd_amd.core.bbox, 'static_assign_and_sample')
wrap(htd_amd.core.bbox, 'batched_max_iou_assign')
wrap(htd_amd.core.bbox, 'batched_random_sample')
wrap(pg, 'pgraph_refine')
rh = model.roi_head
wrap(rh.bbox_head[0], 'loss', 'head0.loss')
wrap(rh.bbox_head[1], 'loss', 'head1.loss')
wrap(rh.glbctx_head, 'forward', 'sfa.forward')
wrap(rh.glbctx_head, 'loss', 'sfa.loss')
wrap(rh.bbox_roi_extractor[0], 'forward', 'roi_extract')
wrap(rh.bbox_roi_extractor[1], 'forward', 'ba_extract')
wrap(model.rpn_head, 'loss', 'rpn.loss')
wrap(model.rpn_head, 'get_bboxes', 'rpn.get_bboxes')
wrap(model, '_parse_losses', 'parse')
wrap(model, 'extract_feat', 'backbone+fpn')
wrap(rh, 'forward_train_static', 'roi_head(all)')
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.train_step(data)
    torch.cuda.synchronize()
ev = prof.events()
ranges = [e for e in ev if e.name.startswith('Q:')]
# map kernels to ranges through their launching CPU op's time: use correlation via e.kernels of cpu events
agg = collections.OrderedDict()
for r in ranges:
    agg.setdefault(r.name, [0, 0.0, 0])
    agg[r.name][2] += 1
cpu_ops = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU and e.kernels]
for e in cpu_ops:
    t = e.time_range.start
    inside = [r for r in ranges if r.time_range.start <= t <= r.time_range.end]
    for r in inside:
        agg[r.name][0] += len(e.kernels)
        agg[r.name][1] += sum(k.duration for k in e.kernels)
tot_k = sum(len(e.kernels) for e in cpu_ops)
print(f'kernels attributed to CPU ops: {tot_k}')
for k, (n, us, calls) in agg.items():
    print(f'{k[2:]:28s} calls {calls:3d}  kernels {n:5d}  device {us / 1e3:8.3f} ms')
