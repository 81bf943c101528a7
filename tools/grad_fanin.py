#!/usr/bin/env python3
"""Autograd nodes of one train step whose output gradient is the sum of several contributions (every extra one is an
`add_` of a full gradient map): node name, number of contributions per output, and the gradient shapes seen."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd.configs import build_htd_detector
from htd_amd.runner import synthetic_batch

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
data = synthetic_batch(4, device=dev)
losses = model.forward_train(**data) if isinstance(data, dict) else model.forward_train(*data)
loss, _ = model._parse_losses(losses)
fan = collections.Counter()
seen, stack = set(), [loss.grad_fn]
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    for nxt, idx in n.next_functions:
        if nxt is not None:
            fan[(nxt, idx)] += 1
            stack.append(nxt)
shapes = {}


def mk(node):
    def hook(gin, gout):
        shapes[node] = [tuple(g.shape) if g is not None else None for g in gout]
    return hook


multi = [(n, i, c) for (n, i), c in fan.items() if c > 1]
for n, i, c in multi:
    n.register_hook(mk(n))
loss.backward()
rows = collections.Counter()
for n, i, c in multi:
    sh = shapes.get(n)
    rows[(n.name(), str(sh[i] if sh and i < len(sh) else '?'), c)] += 1
for (name, sh, c), k in sorted(rows.items(), key=lambda kv: -kv[1]):
    print('%3d x  %-40s output grad %-28s summed from %d contributions' % (k, name, sh, c))
