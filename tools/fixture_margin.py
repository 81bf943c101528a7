#!/usr/bin/env python3
"""How far the product's train-step gradients are from the reference's fixture (tests/golden/detector.npz), in units of the test's
tolerance (tests/test_gpu_detector.py: |diff| <= 2e-4 * max(1, max|ref|) + 1e-3 * |ref| per sampled element), per arithmetic:
the H2 form everywhere it can run, H2 on tensor objects only (no views: the FC stacks and the layers behind strided ones on the
six-product bf16 form), the six-product form everywhere, and the fp32-input matrix instructions.
usage: python tools/fixture_margin.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

import test_gpu_detector as TD
from golden_util import digest, load_seeded_
from htd_amd import capi, dense
from htd_amd.configs import build_htd_detector
from htd_amd.core import set_randperm

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'detector.npz'), allow_pickle=True)
dev = torch.device('cuda:0')
model = build_htd_detector(cfg=TD.small_cfg())
load_seeded_(model, 'det.')
model = model.to(dev).train()
set_randperm(lambda n, device: torch.randperm(n).to(device))
img, metas, gts, labels = TD.inputs(g, dev)
L = capi.lib()
keys = [f[5:-5] for f in g.files if f.startswith('grad.') and f.endswith('.sums')]


def run(name, h2, views, math):
    L.htd_conv2d_set_h2(h2)
    L.htd_conv2d_set_math(math)
    dense.H2_VIEWS = views
    dense.new_step()
    torch.manual_seed(int(g['seed_sampler']))
    losses = model.forward_train(img, metas, gts, labels)
    loss, log_vars = model._parse_losses(losses)
    model.zero_grad()
    loss.backward()
    params = dict(model.named_parameters())
    worst, over, total, rms = [], 0, 0, []
    for k in keys:
        gr = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
        _, sample = digest(gr.detach().cpu())
        ref = g['grad.' + k + '.sample']
        tol = 2e-4 * max(1.0, np.abs(ref).max()) + 1e-3 * np.abs(ref)
        r = np.abs(sample - ref) / tol
        worst.append((float(r.max()), k))
        over += int((r > 1).sum())
        total += r.size
        rms.append(float(np.sqrt(np.mean(r ** 2))))
    worst.sort(reverse=True)
    lossd = max(abs(float(v) - float(g['loss.' + k])) / (1e-4 + 5e-4 * abs(float(g['loss.' + k]))) for k, v in log_vars.items())
    print(f'{name:34s} worst {worst[0][0]:5.2f} x tol ({worst[0][1]}), next {worst[1][0]:4.2f} {worst[2][0]:4.2f}; '
          f'{over} of {total} sampled elements over; rms {np.mean(rms):.3f} x tol (worst tensor {max(rms):.3f}); losses {lossd:4.2f} x tol')


run('H2 wherever a maximum is known', 1, True, 1)
run('H2, tensor objects only', 1, False, 1)
run('six-product bf16 form', 0, True, 1)
run('fp32-input matrix instructions', 0, True, 0)
run('H2 wherever a maximum is known', 1, True, 1)
set_randperm(None)
