"""Time the grouped-conv kernels (csrc/gconv.hip) on the X101-64x4d stage shapes at B=4, 800x1344 input.
usage: python tools/bench_gconv.py [--cols]   (--cols: over DCN column buffers)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htd_amd import dense  # noqa: E402

cols = '--cols' in sys.argv
dev = torch.device('cuda:0')
CL = torch.channels_last
torch.manual_seed(0)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, C, H, W in [('stage1', 256, 200, 336), ('stage2', 512, 100, 168), ('stage3', 1024, 50, 84), ('stage4', 2048, 25, 42)]:
    B, groups = 4, 64
    cg = C // groups
    geom = (C, groups, 3, 3, 1, 1, 1)
    w = (torch.randn(C, cg, 3, 3, device=dev) * 0.05).contiguous(memory_format=CL)
    M = B * H * W
    x = torch.randn(M, 9, C, device=dev) if cols else torch.randn(B, C, H, W, device=dev).contiguous(memory_format=CL)
    g = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=CL)
    wp, wpT = dense._gconv_pack(w, groups, False), dense._gconv_pack(w, groups, True)
    shape = (B, C, H, W)
    flops = 2.0 * M * C * 9 * cg
    t_f = timed(lambda: dense._gconv_fwd_raw(x, wp, None, geom, True, (B, H, W) if cols else None))
    t_d = timed(lambda: dense._gconv_dgrad_raw(g, wpT, geom, shape, cols=cols))
    t_w = timed(lambda: dense._gconv_wgrad_raw(x, g, w, geom, shape, cols=cols))
    byt = 4.0 * (x.numel() + g.numel())
    print(f'{name} C={C} cg={cg} M={M} cols={cols}: fwd {t_f * 1e3:7.1f} us ({flops / t_f / 1e9:6.1f} TF, {byt / t_f / 1e9:5.2f} TB/s)  '
          f'dgrad {t_d * 1e3:7.1f} us ({byt / t_d / 1e9:5.2f} TB/s)  wgrad {t_w * 1e3:7.1f} us ({flops / t_w / 1e9:6.1f} TF, {byt / t_w / 1e9:5.2f} TB/s)')
