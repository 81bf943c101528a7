"""Run one plane-fed 1x1 convolution shape (conv_x3q_kernel) a few times, with the residual epilogue (for rocprofv3 --pmc passes).
usage: one_conv_q.py Ci H W Co [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htd_amd import dense  # noqa: E402

Ci, H, W, Co = [int(v) for v in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dev = torch.device('cuda:0')
CL = torch.channels_last
x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=CL)
w = (torch.randn(Co, Ci, 1, 1, device=dev) * 0.05).contiguous(memory_format=CL)
r = torch.randn(B, Co, H, W, device=dev).contiguous(memory_format=CL)
xp = dense.act_planes(x)
for _ in range(6):
    y = dense._fwd_raw(x, w, None, r, 1, 0, 1, True, x_planes=xp)
torch.cuda.synchronize()
print('algorithmic MB', (x.numel() + w.numel() + 2 * y.numel()) * 4 / 1e6)
