#!/usr/bin/env python3
"""Gradients of one train step with shared-parameter gradients collected in their flat-buffer slice by the kernels
(dense.SINK_ACCUMULATE) against the same step with autograd summing them: per-parameter maximum difference."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import dense
from htd_amd.configs import build_htd_detector, htd_config
from htd_amd.runner import FlatParams, synthetic_batch

dev = torch.device('cuda:0')
cfg = htd_config(50)
cfg.train_cfg.rpn_proposal.update(nms_pre=300, nms_post=200, max_num=200)
for r in cfg.train_cfg.rcnn:
    r.sampler.num = 64
torch.manual_seed(1)
base = build_htd_detector(cfg=cfg).to(dev).train()
data = synthetic_batch(2, 192, 256, 250, device=dev, seed=40)
out = {}
for acc in (False, True):
    dense.SINK_ACCUMULATE = acc
    model = copy.deepcopy(base)
    flat = FlatParams(model, bucket_mb=8)
    for it in range(2):
        flat.zero_grad()
        torch.manual_seed(5)
        model.train_step(data, None)['loss'].backward()
        dense.join_side_stream()
        flat.collect()
    torch.cuda.synchronize()
    out[acc] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    flat.close()
bad = 0
for n in out[True]:
    a, b = out[True][n], out[False][n]
    d = float((a - b).abs().max())
    if d > 1e-6 * max(1e-12, float(b.abs().max())):
        bad += 1
        print(f'{n:60s} max diff {d:.3e}  (|g| max {float(b.abs().max()):.3e})')
print('parameters that differ beyond 1e-6 relative:', bad, 'of', len(out[True]))
