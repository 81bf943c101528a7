"""Run one convolution shape a few times (for rocprofv3 --pmc passes).
usage: one_conv.py Ci H W Co k [B] [fwd|wgrad]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htd_amd import dense  # noqa: E402

Ci, H, W, Co, k = [int(v) for v in sys.argv[1:6]]
B = int(sys.argv[6]) if len(sys.argv) > 6 else 4
dev = torch.device('cuda:0')
CL = torch.channels_last
x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=CL)
w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=CL)
mode = sys.argv[7] if len(sys.argv) > 7 else 'fwd'
dense.tag_amax(x, dense.absmax(x))          # as a tensor written by the package's epilogues: H2 where the kernels have it
y = dense._fwd_raw(x, w, None, None, 1, k // 2, 1, True)
g = torch.randn_like(y)
for _ in range(5):
    if mode == 'wgrad':
        dense._wgrad_raw(x, g, w, 1, k // 2, 1)
    else:
        y = dense._fwd_raw(x, w, None, None, 1, k // 2, 1, True)
torch.cuda.synchronize()
print('algorithmic MB', (x.numel() + w.numel() + y.numel()) * 4 / 1e6)
