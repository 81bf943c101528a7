#!/usr/bin/env python3
"""Weight-gradient entry point (htd_conv2d_bwd_weight) on the layer shapes of the HTD-R50 step: device time per call,
algorithmic TF/s and a checksum of the result bits (two runs with different kernel switches -- HTD_WGRAD_X3D=0/1 --
must print the same checksums: the kernels sum in the same order).

    python tools/bench_wgrad.py [--all] [--only <substring of a layer name>]      (--all: the 3x3 / strided shapes too)
"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import capi, dense

CL = torch.channels_last
POINTWISE = [  # name, B, Ci, H, W, Co, k, stride, pad
    ('l1.conv1 1x1 256-64', 4, 256, 200, 336, 64, 1, 1, 0),
    ('l1.conv3 1x1 64-256', 4, 64, 200, 336, 256, 1, 1, 0),
    ('fpn lat P2 1x1 256-256', 4, 256, 200, 336, 256, 1, 1, 0),
    ('l2.conv1 1x1 512-128', 4, 512, 100, 168, 128, 1, 1, 0),
    ('l2.conv3 1x1 128-512', 4, 128, 100, 168, 512, 1, 1, 0),
    ('fpn lat P3 1x1 512-256', 4, 512, 100, 168, 256, 1, 1, 0),
    ('l3.conv1 1x1 1024-256', 4, 1024, 50, 84, 256, 1, 1, 0),
    ('l3.conv3 1x1 256-1024', 4, 256, 50, 84, 1024, 1, 1, 0),
    ('l4.conv1 1x1 2048-512', 4, 2048, 25, 42, 512, 1, 1, 0),
    ('l4.conv3 1x1 512-2048', 4, 512, 25, 42, 2048, 1, 1, 0),
    ('fc1 12544-1024 x4096', 4096, 12544, 1, 1, 1024, 1, 1, 0),
    ('fc1 12544-1024 x2048', 2048, 12544, 1, 1, 1024, 1, 1, 0),
    ('fc2 1024-1024 x2048', 2048, 1024, 1, 1, 1024, 1, 1, 0),
    ('ragged 1x1 100-132 x1999', 1999, 100, 1, 1, 132, 1, 1, 0),
]
OTHER = [
    ('stem 7x7 s2 8-64', 4, 8, 800, 1344, 64, 7, 2, 3),
    ('l2.0.conv2 3x3 s2 128', 4, 128, 200, 336, 128, 3, 2, 1),
    ('l3.0.conv2 3x3 s2 256', 4, 256, 100, 168, 256, 3, 2, 1),
    ('l4.0.conv2 3x3 s2 512', 4, 512, 50, 84, 512, 3, 2, 1),
    ('l2.0.down 1x1 s2 256-512', 4, 256, 200, 336, 512, 1, 2, 0),
    ('reg 3x3 576 7x7 x24', 24, 576, 7, 7, 576, 3, 1, 1),
    ('l3.conv2 3x3 256', 4, 256, 50, 84, 256, 3, 1, 1),
    ('fpn P2 3x3 256', 4, 256, 200, 336, 256, 3, 1, 1),
    ('fpn P3 3x3 256', 4, 256, 100, 168, 256, 3, 1, 1),
    ('fpn P5 3x3 256', 4, 256, 25, 42, 256, 3, 1, 1),
    ('fpn P6 3x3 256', 4, 256, 13, 21, 256, 3, 1, 1),
    ('l1.conv2 3x3 64', 4, 64, 200, 336, 64, 3, 1, 1),
    ('l2.conv2 3x3 128', 4, 128, 100, 168, 128, 3, 1, 1),
    ('l4.conv2 3x3 512', 4, 512, 25, 42, 512, 3, 1, 1),
    ('ragged 3x3 36-68 9x17 x2', 2, 36, 9, 17, 68, 3, 1, 1),
    ('ragged 3x3 132-200 5x15 x3', 3, 132, 5, 15, 200, 3, 1, 1),
]


def main():
    layers = POINTWISE + (OTHER if '--all' in sys.argv else [])
    if '--only' in sys.argv:
        key = sys.argv[sys.argv.index('--only') + 1]
        layers = [l for l in POINTWISE + OTHER if key in l[0]]
    capi.lib()
    dev = torch.device('cuda', 0)
    print(f'{"layer":34s} {"us":>9s} {"TF/s":>8s}  crc32(gw) crc32(gb)')
    for name, B, Ci, H, W, Co, k, s, p in layers:
        g0 = torch.Generator(device='cpu').manual_seed(1234)
        x = torch.randn(B, Ci, H, W, generator=g0).to(dev).contiguous(memory_format=CL)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        gy = torch.randn(B, Co, Ho, Wo, generator=g0).to(dev).contiguous(memory_format=CL)
        w = torch.empty(Co, Ci, k, k, device=dev).contiguous(memory_format=CL)
        # inside the step both operands carry their maxima (left by the epilogues that wrote them): the same here, so that the table
        # shows the arithmetic the step runs -- H2 where the kernels have it (HTD_CONV_H2=0 / HTD_H2_WGRAD=0: the bf16 form)
        dense.tag_amax(x, dense.absmax(x))
        dense.tag_amax(gy, dense.absmax(gy))
        amax = dense._wgrad_amax(x, gy, w, s, p, 1)
        for it in range(45):                 # 25 launches first: the clocks of an idle device take milliseconds to come up
            if it == 25:
                capi.profile_begin()
            gw, gb = dense._wgrad_launch(x, gy, w, s, p, 1, True, amax)[:2]
        prof = capi.profile_end()
        n, ms = prof['htd_conv2d_bwd_weight_h2' if 'htd_conv2d_bwd_weight_h2' in prof else 'htd_conv2d_bwd_weight'][:2]
        name = name + (' [H2]' if amax is not None else '')
        us = ms / n * 1e3
        flop = 2.0 * B * Ho * Wo * Co * k * k * Ci
        crc_w = zlib.crc32(gw.detach().cpu().contiguous(memory_format=CL).numpy().tobytes())
        crc_b = zlib.crc32(gb.detach().cpu().numpy().tobytes())
        print(f'{name:34s} {us:9.1f} {flop / us / 1e6:8.1f}  {crc_w:08x} {crc_b:08x}')


if __name__ == '__main__':
    main()
