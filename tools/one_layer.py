#!/usr/bin/env python3
"""Run one conv layer's fwd / dgrad / wgrad a few times (for rocprofv3 counter collection)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import dense

CL = torch.channels_last
B, Ci, H, W, Co, k, s, p = [int(v) for v in (sys.argv[1:9] if len(sys.argv) > 8 else (4, 256, 200, 336, 256, 3, 1, 1))]
dev = torch.device('cuda:0')
x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=CL).requires_grad_()
w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5).contiguous(memory_format=CL).requires_grad_()
for _ in range(3):
    y = dense.conv2d(x, w, None, s, p, 1)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
