#!/usr/bin/env python3
"""bf16 forward convolution (htd_conv2d_fwd_bf16) on the HTD-R50 layer shapes, B = 4 @ 800x1344: TFLOP/s from device
events, next to ATen's bf16 convolution."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from htd_amd import capi, dense

CL = torch.channels_last
LAYERS = [('l2.conv2 3x3 128', 128, 100, 168, 128, 3, 1, 1), ('l3.conv2 3x3 256', 256, 50, 84, 256, 3, 1, 1),
          ('l3.conv3 1x1 256-1024', 256, 50, 84, 1024, 1, 1, 0), ('l4.conv2 3x3 512', 512, 25, 42, 512, 3, 1, 1),
          ('fpn P2 3x3 256', 256, 200, 336, 256, 3, 1, 1), ('fpn P3 3x3 256', 256, 100, 168, 256, 3, 1, 1)]
dev = torch.device('cuda:0')
print(f'{"layer":24s} {"GFLOP":>8s} | {"fwd":>7s} {"wgrad":>7s} TF/s (htd bf16) | {"ATen bf16 fwd":>13s}')
for name, Ci, H, W, Co, k, s, p in LAYERS:
    x = torch.randn(4, Ci, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5).to(torch.bfloat16).contiguous(memory_format=CL)
    y = dense.conv2d_bf16(x, w, None, s, p, 1)
    flop = 2.0 * y.numel() * Ci * k * k
    for it in range(12):
        if it == 2:
            capi.profile_begin()
        dense.conv2d_bf16(x, w, None, s, p, 1)
    calls, ms = capi.profile_end()['htd_conv2d_fwd_bf16'][:2]
    gy = torch.randn_like(y.float()).to(torch.bfloat16).contiguous(memory_format=CL)
    for it in range(12):
        if it == 2:
            capi.profile_begin()
        dense.conv2d_wgrad_bf16(x, gy, w.shape, s, p, 1)
    wcalls, wms = capi.profile_end()['htd_conv2d_bwd_weight_bf16'][:2]
    F.conv2d(x, w, None, s, p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        F.conv2d(x, w, None, s, p)
    torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / 10
    print(f'{name:24s} {flop / 1e9:8.1f} | {flop / (ms / calls * 1e-3) / 1e12:7.1f} {flop / (wms / wcalls * 1e-3) / 1e12:7.1f}'
          f'                   | {flop / ta / 1e12:13.1f}')
