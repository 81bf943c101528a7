#!/usr/bin/env python3
"""conv_x3q_kernel (both operands pre-split, LDS-DMA only) against conv_x3p_kernel (activations split in the K loop) on the 1x1
layer shapes of HTD-R50 / R101 @ B=4, 800x1344, forward and data-gradient form, and what the producers pay for writing the planes
(3x3 layer with / without `emit`, htd_act_planes as a pass of its own).  Run with HTD_X3P_TUNE=1 (re-reads the variant switches
per call).  usage: HTD_X3P_TUNE=1 python tools/bench_planes.py [substring of layer name]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import capi, dense

CL = torch.channels_last
ONE = [  # name, Ci, H, W, Co  (1x1, stride 1)
    ('l1.conv3 64-256', 64, 200, 336, 256),
    ('l1.conv1 256-64', 256, 200, 336, 64),
    ('l2.conv3 128-512', 128, 100, 168, 512),
    ('l2.conv1 512-128', 512, 100, 168, 128),
    ('l3.conv3 256-1024', 256, 50, 84, 1024),
    ('l3.conv1 1024-256', 1024, 50, 84, 256),
    ('l4.conv3 512-2048', 512, 25, 42, 2048),
    ('l4.conv1 2048-512', 2048, 25, 42, 512),
    ('fpn lat P3 512-256', 512, 100, 168, 256),
    ('fpn lat P4 1024-256', 1024, 50, 84, 256),
]
THREE = [  # producers: 3x3 s1
    ('l2.conv2 3x3 128', 128, 100, 168),
    ('l3.conv2 3x3 256', 256, 50, 84),
    ('l4.conv2 3x3 512', 512, 25, 42),
]


def timed(fn, key, n=10, warm=20):
    for _ in range(warm):
        fn()
    capi.profile_begin()
    for _ in range(n):
        fn()
    prof = capi.profile_end()
    calls, ms = prof[key][0], prof[key][1]
    return ms / calls * 1e3          # us per call


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    dev = torch.device('cuda:0')
    B = 4
    tune = os.environ.get('HTD_X3P_TUNE') is not None
    variants = [(32, 2), (16, 2), (32, 3), (16, 3)] if tune else [(0, 0)]
    print(f'{"layer":24s} {"GFLOP":>6s} | x3p us (TF/s) | ' + ' | '.join(f'x3q mf{m} ns{n}' for m, n in variants) + ' | bit-equal')
    for name, Ci, H, W, Co in ONE:
        if only and only not in name:
            continue
        x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(Co, Ci, 1, 1, device=dev) / Ci ** 0.5).contiguous(memory_format=CL)
        r = torch.randn(B, Co, H, W, device=dev).contiguous(memory_format=CL)
        flop = 2.0 * B * H * W * Ci * Co
        xp = dense.act_planes(x)
        for label, res in (('', None), (' +res', r)):
            t0 = timed(lambda: dense._fwd_raw(x, w, None, res, 1, 0, 1, res is not None), 'htd_conv2d_fwd_x3p')
            y0 = dense._fwd_raw(x, w, None, res, 1, 0, 1, res is not None)
            cells, same = [], True
            for mf, ns in variants:
                if tune:
                    os.environ['HTD_X3Q_MFMA'], os.environ['HTD_X3Q_NS'] = str(mf), str(ns)
                t = timed(lambda: dense._fwd_raw(x, w, None, res, 1, 0, 1, res is not None, x_planes=xp), 'htd_conv2d_fwd_x3p')
                y = dense._fwd_raw(x, w, None, res, 1, 0, 1, res is not None, x_planes=xp)
                same = same and (bool(torch.equal(y, y0)) or mf == 32)         # (x3p's default 1x1 shape is 16x16x32)
                cells.append(f'{t:7.1f} ({flop / t / 1e6:5.1f})')
            print(f'{name + label:24s} {flop / 1e9:6.1f} | {t0:7.1f} ({flop / t0 / 1e6:5.1f}) | ' + ' | '.join(cells) + f' | {same}')
    print()
    print(f'{"producer":24s} | plain us | emit us | act_planes pass us')
    for name, C, H, W in THREE:
        if only and only not in name:
            continue
        x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(C, C, 3, 3, device=dev) / (9 * C) ** 0.5).contiguous(memory_format=CL)
        t0 = timed(lambda: dense._fwd_raw(x, w, None, None, 1, 1, 1, True), 'htd_conv2d_fwd_x3p')
        t1 = timed(lambda: dense._fwd_raw(x, w, None, None, 1, 1, 1, True, emit=True), 'htd_conv2d_fwd_x3p')
        t2 = timed(lambda: dense.act_planes(x), 'htd_act_planes')
        print(f'{name:24s} | {t0:8.1f} | {t1:7.1f} | {t2:8.1f}')


if __name__ == '__main__':
    main()
