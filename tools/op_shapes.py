#!/usr/bin/env python3
"""Top aten ops of one train step by device time with their input shapes (what is left outside the C ABI)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_step(data)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and e.name.startswith('aten::'):
        key = (e.name, str(e.input_shapes)[:90])
        agg[key][0] += 1
        agg[key][1] += sum(k.duration for k in e.kernels)
want = sys.argv[1] if len(sys.argv) > 1 else None
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print(f'aten ops with kernels: {sum(v[0] for v in agg.values())} calls, {tot / 1e3:.2f} ms device')
for (name, shp), (n, us) in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    if want and want not in name:
        continue
    print(f'{us / 1e3:7.3f} ms  n={n:3d}  {name:28s} {shp}')
