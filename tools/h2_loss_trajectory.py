#!/usr/bin/env python3
"""The same training run (same initial weights, same four synthetic batches, same sampling seeds) on the H2 arithmetic and on the
three-piece bf16 form: total loss per step side by side.  Both are fp32-accurate, not bit-equal, and a detector's training
amplifies rounding (sample flips at IoU thresholds): the curves agree to ~1e-5 at first and drift apart like two fp32 runs with
different summation orders do.  usage: python tools/h2_loss_trajectory.py [steps] [lr]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import capi, dense
from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
dev = torch.device('cuda:0')
L = capi.lib()
curves = {}
for mode in ('bf16x6', 'h2', 'bf16x6 again'):
    L.htd_conv2d_set_h2(1 if mode == 'h2' else 0)
    dense.new_step()
    torch.manual_seed(0)
    model = build_htd_detector(50).to(dev).train()
    tr = Trainer(model, lr=lr)
    datas = [synthetic_batch(4, device=dev, seed=s) for s in range(4)]
    losses = []
    for i in range(steps):
        torch.manual_seed(1000 + i)
        out = tr.train_step(datas[i % 4])
        losses.append(float(out['loss']))
    curves[mode] = losses
    if mode == 'h2':
        dense.h2_check()
L.htd_conv2d_set_h2(1)
a, b, c = curves['bf16x6'], curves['h2'], curves['bf16x6 again']
print('step   bf16x6        h2            rel.diff     (bf16x6 run twice: rel.diff)')
for i in range(steps):
    if i < 12 or i % 10 == 0 or i == steps - 1:
        print(f'{i:4d}  {a[i]:12.6f}  {b[i]:12.6f}  {abs(a[i] - b[i]) / abs(a[i]):10.2e}   {abs(a[i] - c[i]) / abs(a[i]):10.2e}')
fin = all(x == x and abs(x) != float('inf') for x in b)
print('all H2 losses finite:', fin, ' mean of the last 10:', sum(a[-10:]) / 10, sum(b[-10:]) / 10)
sys.exit(0 if fin else 1)
