#!/usr/bin/env python3
"""Which convolution launches of a train step found no carried maximum on their input (dense.carried_amax) -- and so took a pass of
htd_absmax or stayed on the three-piece bf16 form.  One line per (kind, tensor shape, weight shape, provenance)."""
import os, sys
os.environ['HTD_H2_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from htd_amd import dense
from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
for _ in range(2):
    tr.train_step(data)
dense.H2_TRACE.clear()
tr.train_step(data)
torch.cuda.synchronize()
for k, v in sorted(dense.H2_TRACE.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(v, k)
