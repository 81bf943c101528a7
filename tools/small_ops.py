#!/usr/bin/env python3
"""Which tensors the small ATen kernels of one train step work on: add / fill / copy ops grouped by shape."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.train_step(data)
    torch.cuda.synchronize()
want = ('aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::zeros', 'aten::clone', 'aten::contiguous',
        'aten::mul', 'aten::sum', 'aten::cat', 'aten::index_select', 'aten::to', 'aten::_to_copy')
c = collections.Counter()
for e in prof.events():
    if e.name in want and (e.cpu_parent is None or not e.cpu_parent.name.startswith('aten::')):
        par = e.cpu_parent.name if e.cpu_parent is not None else '-'
        c[(e.name, str(e.input_shapes)[:70], par[:50])] += 1
for (n, sh, par), k in c.most_common(70):
    print(f'{k:4d} {n:18s} {sh:70s} <- {par}')
