"""Time htd_deform_col2im on one layer shape.  usage: python tools/bench_col2im.py B H W C [offset_std]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htd_amd import capi  # noqa: E402

B, H, W, C = [int(v) for v in sys.argv[1:5]]
std = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
dev = torch.device('cuda:0')
torch.manual_seed(0)
x = torch.randn(B, H, W, C, device=dev)
off = torch.randn(B, H, W, 18, device=dev) * std
gcol = torch.randn(B * H * W, 9, C, device=dev)
gx = torch.zeros_like(x)
goff = torch.empty_like(off)
P, S = capi.ptr, capi.current_stream_ptr


MODE = os.environ.get('C2I_MODE', 'all')


def run():
    capi.call('htd_deform_col2im', P(x), P(off), None, P(gcol), None if MODE == 'goff' else P(gx), None if MODE == 'gx' else P(goff), None, B, H, W, C, 3, 3, 1, 1, 1, 1, S())


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    run()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(f'col2im B{B} {H}x{W} C{C} std {std} mode={MODE} direct={os.environ.get("HTD_DCN_DIRECT_COL2IM")}: '
      f'{ms:.3f} ms  (gcol {gcol.numel() * 4 / 1e6:.0f} MB -> {gcol.numel() * 4 / ms / 1e6:.0f} GB/s)')
