#!/usr/bin/env python3
"""Who launches the big ATen element-wise / copy kernels of one train step?  A TorchDispatchMode logs every aten op whose
largest tensor argument has >= MIN_NUMEL elements, with the innermost package frames of the Python stack (ops issued by the
C++ autograd engine show the frame that called backward()).  usage: python tools/aten_trace.py [min_numel] [--bf16]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

ARGS = [a for a in sys.argv[1:] if not a.startswith('--')]
MIN = int(ARGS[0]) if ARGS else 1_000_000
SKIP = ('aten.view', 'aten._unsafe_view', 'aten.permute', 'aten.detach', 'aten.alias', 'aten.slice', 'aten.select', 'aten.t.',
        'aten.transpose', 'aten.as_strided', 'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.reshape', 'aten.empty', 'aten.new_empty',
        'aten.empty_like', 'aten.narrow', 'aten.split', 'aten.unbind', 'aten._reshape_alias', 'aten.is_', 'aten.record_stream', 'aten.lift_fresh')
log = collections.Counter()


class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            big = max([a.numel() for a in tree_flatten((args, kwargs or {}))[0] if isinstance(a, torch.Tensor)] or [0])
            if big >= MIN:
                shapes = [tuple(a.shape) for a in tree_flatten((args, kwargs or {}))[0] if isinstance(a, torch.Tensor)][:3]
                frames = [f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in traceback.extract_stack()
                          if 'htd_amd' in f.filename and 'aten_trace' not in f.filename][-3:]
                log[(name, str(shapes), ' < '.join(reversed(frames)))] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device('cuda:0')
torch.manual_seed(0)
if '--bf16' in sys.argv:            # BASELINE configs[2] per-GPU shape
    model = build_htd_detector(101, bf16=True).to(dev).train()
    tr = Trainer(model, lr=0.015, comm_dtype=torch.bfloat16)
else:
    model = build_htd_detector(50).to(dev).train()
    tr = Trainer(model)
data = synthetic_batch(4, device=dev)
for _ in range(2):
    tr.train_step(data)
torch.cuda.synchronize()
with Trace():
    tr.train_step(data)
torch.cuda.synchronize()
if '--sites' in sys.argv:        # launches per call site, whatever the shapes
    sites = collections.Counter()
    for (name, shapes, where), n in log.items():
        sites[(where or '(autograd engine)', name)] += n
    for (where, name), n in sorted(sites.items(), key=lambda kv: -kv[1])[:int(os.environ.get('TRACE_TOP', '80'))]:
        print(f'{n:3d}  {name:34s} {where}')
    print('total', sum(sites.values()))
else:
    for (name, shapes, where), n in sorted(log.items(), key=lambda kv: -kv[1]):
        print(f'{n:3d}  {name:34s} {shapes[:70]:70s} {where}')
