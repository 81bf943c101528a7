#!/usr/bin/env python3
"""bf16 forward convolution (htd_conv2d_fwd_bf16) on the HTD-R50 layer shapes, B = 4 @ 800x1344: TFLOP/s from device
events, next to ATen's bf16 convolution."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from htd_amd import capi, dense

CL = torch.channels_last
LAYERS = [('l2.conv2 3x3 128', 128, 100, 168, 128, 3, 1, 1), ('l2.conv3 1x1 128-512', 128, 100, 168, 512, 1, 1, 0),
          ('l2.conv1 1x1 512-128', 512, 100, 168, 128, 1, 1, 0), ('l3.conv2 3x3 256', 256, 50, 84, 256, 3, 1, 1),
          ('l3.conv3 1x1 256-1024', 256, 50, 84, 1024, 1, 1, 0), ('l3.conv1 1x1 1024-256', 1024, 50, 84, 256, 1, 1, 0),
          ('l4.conv2 3x3 512', 512, 25, 42, 512, 3, 1, 1), ('l4.conv3 1x1 512-2048', 512, 25, 42, 2048, 1, 1, 0),
          ('l4.conv1 1x1 2048-512', 2048, 25, 42, 512, 1, 1, 0),
          ('fpn P2 3x3 256', 256, 200, 336, 256, 3, 1, 1), ('fpn P3 3x3 256', 256, 100, 168, 256, 3, 1, 1),
          ('fc1 12544-1024 x2048', 12544, 2048, 1, 1024, 1, 1, 0)]
# with HTD_BF16Q_TUNE=1: conv_bf16_kernel (HTD_BF16Q=0) against every tile / ring depth of conv_bf16q_kernel, us per call
VARIANTS = [('old', None, None), ('q64 ns2', 64, 2), ('q64 ns3', 64, 3), ('q64 ns4', 64, 4), ('q128 ns2', 128, 2), ('q128 ns3', 128, 3)]
if os.environ.get('BENCH_ABLATE'):      # HTD_BF16Q_DBG: 1 = no epilogue, 2 = no K loop (results are wrong; only the time matters)
    VARIANTS = [('old', None, None), ('q64 ns2', 64, 2), ('q64 ns3', 64, 3), ('q64 ns4', 64, 4), ('q128 ns2', 128, 2), ('q128 ns3', 128, 3)]
dev = torch.device('cuda:0')
if os.environ.get('HTD_BF16Q_TUNE'):
    print(f'{"layer":24s} {"GFLOP":>7s} {"MB":>6s} | ' + ' | '.join(f'{v[0]:>9s}' for v in VARIANTS) + '   us per call (TF/s of the best)')
    for name, Ci, H, W, Co, k, s, p in LAYERS:
        if len(sys.argv) > 1 and sys.argv[1] not in name:
            continue
        B = 1 if name.startswith('fc1') else 4
        x = torch.randn(B, Ci, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
        w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5).to(torch.bfloat16).contiguous(memory_format=CL)
        y = dense.conv2d_bf16(x, w, None, s, p, 1)
        flop = 2.0 * y.numel() * Ci * k * k
        mb = 2.0 * (x.numel() + w.numel() + y.numel()) / 1e6
        cells = []
        for label, tile, ns in VARIANTS:
            if ns == 4 and k == 3:
                cells.append(None)
                continue
            os.environ['HTD_BF16Q'] = '0' if tile is None else '1'
            if tile is not None:
                os.environ['HTD_BF16Q_TILE'], os.environ['HTD_BF16Q_NS'] = str(tile), str(ns)
            for it in range(30):
                if it == 20:
                    capi.profile_begin()
                dense.conv2d_bf16(x, w, None, s, p, 1)
            calls, ms = capi.profile_end()['htd_conv2d_fwd_bf16'][:2]
            cells.append(ms / calls * 1e3)
        best = min(c for c in cells if c is not None)
        print(f'{name:24s} {flop / 1e9:7.1f} {mb:6.1f} | ' + ' | '.join('        -' if c is None else f'{c:9.1f}' for c in cells) +
              f'   ({flop / best / 1e6:6.1f} TF/s, {mb / best * 1e-3:5.2f} TB/s)')
    sys.exit(0)
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
