#!/usr/bin/env python3
"""Which convolution launches of one train step run on which arithmetic, and what each costs: the matrix-core entry points timed
per layer shape (capi detail mode, weight gradients on the main stream so that no two launches share the device).
usage: python tools/conv_layers.py [depth] [rows]"""
import os
import sys

os.environ.setdefault('HTD_OVERLAP_WGRAD', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import capi
from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

ENTRY = ('htd_conv2d_fwd_x3h', 'htd_conv2d_bwd_data_x3h', 'htd_conv2d_bwd_weight_h2', 'htd_conv2d_fwd_x3p', 'htd_conv2d_bwd_data_x3p',
         'htd_conv2d_fwd', 'htd_conv2d_bwd_data', 'htd_conv2d_bwd_weight', 'htd_conv2d_stem7_fwd')
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(int(sys.argv[1]) if len(sys.argv) > 1 else 50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
for _ in range(4):
    tr.train_step(data)
torch.cuda.synchronize()
N = 5
capi.profile_begin(detail=True, only=ENTRY)
for _ in range(N):
    tr.train_step(data)
prof = capi.profile_end()
rows = sorted(prof.items(), key=lambda kv: -kv[1][1])
by_entry = {}
for k, (calls, ms, kind, work, nbytes) in rows:
    e = k.split('(')[0]
    c = by_entry.setdefault(e, [0, 0.0, 0.0])
    c[0] += calls / N
    c[1] += ms / N
    c[2] += work / N
print(f'{"entry point":30s} {"calls":>6s} {"ms/step":>8s} {"TF/s":>7s}')
for e, (c, ms, w) in sorted(by_entry.items(), key=lambda kv: -kv[1][1]):
    print(f'{e:30s} {c:6.0f} {ms:8.3f} {w / ms / 1e9 if ms else 0:7.1f}')
print()
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 70
print(f'{"ms/step":>8s} {"calls":>5s} {"TF/s":>7s}  launch (integer arguments)[pointer operands]')
for k, (calls, ms, kind, work, nbytes) in rows[:lim]:
    print(f'{ms / N:8.3f} {calls / N:5.0f} {work / ms / 1e9 if ms else 0:7.1f}  {k}')
