#!/usr/bin/env python3
"""cProfile of the host side of the headline train step: where the ~28 ms of issue time per step go (Python frames by own time).
usage: python tools/host_profile.py [steps] [--bf16]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
dev = torch.device('cuda:0')
torch.manual_seed(0)
bf16 = '--bf16' in sys.argv
model = build_htd_detector(101 if bf16 else 50, bf16=bf16).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
for _ in range(5):
    tr.train_step(data)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    tr.train_step(data)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime')
print('per-step figures = totals / %d' % steps)
st.print_stats(45)
