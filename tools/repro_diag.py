#!/usr/bin/env python3
"""Is one train step reproducible bit for bit?  Runs the SAME step (same weights, same batch, same sampling keys) several times
and lists the parameters whose gradients differ between runs, with the size of the difference -- the names point at the kernels
that still sum with float atomics.  usage: python tools/repro_diag.py [runs]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd.configs import build_htd_detector, htd_config
from htd_amd.core.bbox import set_sample_keys
from htd_amd.runner import Trainer, synthetic_batch


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device('cuda:0')
    cfg = htd_config(50)
    if os.environ.get('REPRO_SMALL', '1') == '1':          # the configuration of tests/test_gpu_distributed.py
        cfg.train_cfg.rpn_proposal.update(nms_pre=300, nms_post=200, max_num=200)
        for r in cfg.train_cfg.rcnn:
            r.sampler.num = 64
        data = synthetic_batch(2, 192, 256, 250, device=dev, seed=40)
    else:
        data = synthetic_batch(4, 800, 1344, 1333, device=dev, seed=0)
    coef = torch.tensor([12.9898, 78.233, 37.719, 93.989], device=dev)
    set_sample_keys(lambda cand: torch.frac(torch.sin((cand * coef).sum(-1)) * 43758.5453).abs())
    torch.manual_seed(1)
    model0 = build_htd_detector(cfg=cfg).to(dev).train()
    grads = []
    for r in range(runs):
        model = copy.deepcopy(model0)
        tr = Trainer(model, lr=0.0)                      # lr 0: the weights stay, the flat gradient buffer is what we read
        tr.train_step(data)
        torch.cuda.synchronize()
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        grads.append({n: p.grad.detach().clone() if p.grad is not None else None for n, p in model.named_parameters() if p.requires_grad})
        flat = tr.flat.grad.clone()
        if r == 0:
            flat0 = flat
        else:
            d = (flat - flat0).abs()
            print(f'run {r}: flat gradient differs in {int((d > 0).sum())} of {d.numel()} elements, max |diff| {float(d.max()):.3e} '
                  f'(max |grad| {float(flat0.abs().max()):.3e})')
        tr.flat.close()
    bad = []
    for n in names:
        a, b = grads[0][n], grads[1][n]
        if a is None or b is None:
            continue
        d = float((a - b).abs().max())
        if d > 0:
            bad.append((d / max(float(a.abs().max()), 1e-30), d, n))
    bad.sort(reverse=True)
    print(f'{len(bad)} of {len(names)} parameters differ between run 0 and run 1')
    for rel, d, n in bad[:40]:
        print(f'  {rel:9.2e} rel  {d:9.2e} abs  {n}')


if __name__ == '__main__':
    main()
