#!/usr/bin/env python3
"""Forward TFLOP/s of conv_igemm_kernel per tile configuration (HTD_CONV_FORCE_TILE, see conv_fwd.hip) on the
mid-size layer shapes of HTD-R50 / R101 @ B=4, 800x1344.  HTD_CONV_TUNE=1 must be in the environment at load time."""
import os
import sys

os.environ['HTD_CONV_TUNE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import capi, dense

CL = torch.channels_last
LAYERS = [  # name, Ci, H, W, Co, k, stride, pad
    ('l2.conv2 3x3 128', 128, 100, 168, 128, 3, 1, 1),
    ('l2.conv1 1x1 512-128', 512, 100, 168, 128, 1, 1, 0),
    ('l2.conv3 1x1 128-512', 128, 100, 168, 512, 1, 1, 0),
    ('l3.conv2 3x3 256', 256, 50, 84, 256, 3, 1, 1),
    ('l3.conv1 1x1 1024-256', 1024, 50, 84, 256, 1, 1, 0),
    ('l3.conv3 1x1 256-1024', 256, 50, 84, 1024, 1, 1, 0),
    ('l4.conv2 3x3 512', 512, 25, 42, 512, 3, 1, 1),
    ('l4.conv1 1x1 2048-512', 2048, 25, 42, 512, 1, 1, 0),
    ('l4.conv3 1x1 512-2048', 512, 25, 42, 2048, 1, 1, 0),
    ('fpn P3 3x3 256', 256, 100, 168, 256, 3, 1, 1),
    ('fpn P4 3x3 256', 256, 50, 84, 256, 3, 1, 1),
    ('fpn P2 3x3 256', 256, 200, 336, 256, 3, 1, 1),
    ('l1.conv2 3x3 64', 64, 200, 336, 64, 3, 1, 1),
    ('l1.conv3 1x1 64-256', 64, 200, 336, 256, 1, 1, 0),
    ('reg conv 3x3 576 n=512', 576, 7 * 16, 7 * 32, 576, 3, 1, 1),
    ('reg conv 3x3 256-576', 256, 7 * 16, 7 * 32, 576, 3, 1, 1),
    ('reg conv 3x3 576-1024', 576, 7 * 16, 7 * 32, 1024, 3, 1, 1),
    ('reg dgrad 3x3 1024-576', 1024, 7 * 16, 7 * 32, 576, 3, 1, 1),
    ('reg conv 576 n=24', 576, 7 * 6, 7 * 1, 576, 3, 1, 1),
    ('fc 12544-1024 M=2048', 12544, 32, 16, 1024, 1, 1, 0),
    ('fc 1024-1024 M=2048', 1024, 32, 16, 1024, 1, 1, 0),
    ('fc dgrad 1024-12544', 1024, 32, 16, 12544, 1, 1, 0),
    ('rpn P3 3x3 256', 256, 100, 168, 256, 3, 1, 1),
    ('fpn lat 1x1 2048-256', 2048, 25, 42, 256, 1, 1, 0),
    ('fpn lat 1x1 512-256', 512, 100, 168, 256, 1, 1, 0),
    ('l3 ds 1x1 512-1024 s2', 512, 100, 168, 1024, 1, 2, 0),
]
NAMES = ['auto', '64x64', '128x32', '128x64/4x1', '128x128', '128x64/2x2', '64x128']


def rate(x, w, s, p, flop, n=6):
    for it in range(n + 2):
        if it == 2:
            capi.profile_begin()
        dense.conv2d(x, w, None, s, p, 1)
    calls, ms = capi.profile_end()['htd_conv2d_fwd'][:2]
    return flop / (ms / calls * 1e-3) / 1e12


def main():
    dev = torch.device('cuda:0')
    only = sys.argv[1] if len(sys.argv) > 1 else None
    print(f'{"layer":26s} {"GFLOP":>7s} | ' + ' '.join(f'{n:>11s}' for n in NAMES))
    with torch.no_grad():
        for name, Ci, H, W, Co, k, s, p in LAYERS:
            if only and only not in name:
                continue
            x = torch.randn(4, Ci, H, W, device=dev).contiguous(memory_format=CL)
            w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5).contiguous(memory_format=CL)
            y = dense.conv2d(x, w, None, s, p, 1)
            flop = 2.0 * y.numel() * Ci * k * k
            out = []
            for cfg in range(-1, 6):
                os.environ['HTD_CONV_FORCE_TILE'] = str(cfg)
                out.append(rate(x, w, s, p, flop))
            os.environ['HTD_CONV_FORCE_TILE'] = '-1'
            print(f'{name:26s} {flop / 1e9:7.1f} | ' + ' '.join(f'{r:11.1f}' for r in out) +
                  f' | auto/best {out[0] / max(out[1:]):.3f}', flush=True)


if __name__ == '__main__':
    main()
