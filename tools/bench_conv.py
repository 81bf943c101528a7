#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA conv kernels on the layer shapes of HTD-R50 @ B=4, 800x1344
(fwd / data-grad / weight-grad TFLOP/s), next to ATen's convolution for reference."""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from htd_amd import capi, dense

CL = torch.channels_last
LAYERS = [  # name, Ci, H, W, Co, k, stride, pad
    ('l1.conv2 3x3 64', 64, 200, 336, 64, 3, 1, 1),
    ('l1.conv3 1x1 64-256', 64, 200, 336, 256, 1, 1, 0),
    ('l2.conv2 3x3 128', 128, 100, 168, 128, 3, 1, 1),
    ('l2.conv3 1x1 128-512', 128, 100, 168, 512, 1, 1, 0),
    ('l3.conv2 3x3 256', 256, 50, 84, 256, 3, 1, 1),
    ('l3.conv1 1x1 1024-256', 1024, 50, 84, 256, 1, 1, 0),
    ('l4.conv2 3x3 512', 512, 25, 42, 512, 3, 1, 1),
    ('fpn P2 3x3 256', 256, 200, 336, 256, 3, 1, 1),
    ('rpn P2 1x1 256-15', 256, 200, 336, 15, 1, 1, 0),
    ('l2.0.conv2 3x3 s2', 128, 200, 336, 128, 3, 2, 1),
    ('fpn P3 3x3 256', 256, 100, 168, 256, 3, 1, 1),
    ('fpn P5 3x3 256', 256, 25, 42, 256, 3, 1, 1),
    ('fpn P6 3x3 256', 256, 13, 21, 256, 3, 1, 1),
    ('l3.conv3 1x1 256-1024', 256, 50, 84, 1024, 1, 1, 0),
    ('l4.conv3 1x1 512-2048', 512, 25, 42, 2048, 1, 1, 0),
    # the RoI heads' FC layers as 1x1 convolutions over (rows, C, 1, 1): last element = rows
    ('fc 12544-1024 x2048', 12544, 1, 1, 1024, 1, 1, 0, 2048),
    ('fc 12544-1024 x4096', 12544, 1, 1, 1024, 1, 1, 0, 4096),
    ('fc 1024-1024 x2048', 1024, 1, 1, 1024, 1, 1, 0, 2048),
    ('fc 1024-1024 x4096', 1024, 1, 1, 1024, 1, 1, 0, 4096),
]


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def device_rates(x, w, g, s, p, flop, n=8):
    """TF/s of the three C-ABI entry points from device events around each call (capi profile)."""
    xg, wg = x.clone().requires_grad_(), w.clone().requires_grad_()
    if capi.lib().htd_conv2d_set_h2(-1) == 1 and x.dtype == torch.float32:
        # inside the step a layer's input and gradient carry their maxima (left by the epilogues that wrote them): the same here, so
        # that the table shows the arithmetic the step runs (H2 where the kernels have it)
        g = g.clone()
        dense.tag_amax(xg, dense.absmax(xg))
        dense.tag_amax(g, dense.absmax(g))
    for it in range(n + 20):                 # 20 untimed iterations first: an idle device needs milliseconds to raise its clocks
        if it == 20:
            capi.profile_begin()
        dense.conv2d(xg, wg, None, s, p, 1).backward(g)
    prof = capi.profile_end()
    out = []
    for ks in (('htd_conv2d_fwd_x3h', 'htd_conv2d_fwd_x3p', 'htd_conv2d_fwd'), ('htd_conv2d_bwd_data_x3h', 'htd_conv2d_bwd_data_x3p', 'htd_conv2d_bwd_data'), ('htd_conv2d_bwd_weight_h2', 'htd_conv2d_bwd_weight')):
        k = next(k for k in ks if k in prof)          # conv_x3p_kernel where it takes the layer, conv_igemm_kernel otherwise
        calls = max(prof[k][0] for k in ks if k in prof)
        ms = sum(prof[k][1] for k in ks if k in prof)
        out.append(flop / (ms / calls * 1e-3) / 1e12)
    return out


def main():
    B = 4
    dev = torch.device('cuda:0')
    only = sys.argv[1] if len(sys.argv) > 1 else None
    print(f'{"layer":26s} {"GFLOP":>8s} | {"fwd":>7s} {"dgrad":>7s} {"wgrad":>7s} TF/s (htd) | {"fwd":>7s} {"bwd":>7s} TF/s (ATen)')
    for name, Ci, H, W, Co, k, s, p, *rows in LAYERS:
        if only and only not in name:
            continue
        B = rows[0] if rows else 4
        x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5).contiguous(memory_format=CL)
        y = dense.conv2d(x, w, None, s, p, 1)
        g = torch.randn_like(y)
        flop = 2.0 * y.numel() * Ci * k * k
        r_f, r_x, r_w = device_rates(x, w, g, s, p, flop)
        xg = x.clone().requires_grad_()
        wg = w.clone().requires_grad_()
        t_af = timeit(lambda: F.conv2d(x, w, None, s, p))

        def aten_bwd():
            yy = F.conv2d(xg, wg, None, s, p)
            yy.backward(g)
        t_ab = timeit(aten_bwd) - t_af
        print(f'{name:26s} {flop / 1e9:8.1f} | {r_f:7.1f} {r_x:7.1f} {r_w:7.1f}'
              f'             | {flop / t_af / 1e12:7.1f} {2 * flop / t_ab / 1e12:7.1f}')


if __name__ == '__main__':
    main()
