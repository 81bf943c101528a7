#!/usr/bin/env python3
"""Count the operators / kernel launches of one production train step (torch.profiler, CPU side), grouped by the
python frame that issued them.  usage: python tools/step_ops.py [depth]"""
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
model = build_htd_detector(int(sys.argv[1]) if len(sys.argv) > 1 else 50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.train_step(data)
    torch.cuda.synchronize()
ev = prof.events()
launch = [e for e in ev if 'LaunchKernel' in e.name or e.name in ('hipMemcpyAsync', 'hipMemsetAsync', 'hipMemcpyWithStream')]
print('launches', len(launch))
ops = [e for e in ev if e.name.startswith('aten::') and e.cpu_parent is not None and not e.cpu_parent.name.startswith('aten::')]
by_frame = collections.Counter()
by_frame_ops = collections.defaultdict(collections.Counter)
for e in ops:
    frame = next((f for f in (e.stack or []) if '/htd_amd/' in f and 'capi.py' not in f), None)
    if frame is None:
        frame = 'autograd/other: ' + (e.cpu_parent.name if e.cpu_parent is not None else '?')
    else:
        frame = frame.split('/htd_amd/')[-1]
    by_frame[frame] += 1
    by_frame_ops[frame][e.name] += 1
print('top-level aten ops', len(ops))
for f, n in by_frame.most_common(110):
    print(f'{n:5d}  {f[:90]:90s} ' + ', '.join(f'{k[6:]}x{v}' for k, v in by_frame_ops[f].most_common(6)))
