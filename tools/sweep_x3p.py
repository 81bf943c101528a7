#!/usr/bin/env python3
"""conv_x3p_kernel (csrc/conv_x3.hip: pre-split weights, halo-staged activations) against conv_igemm_kernel<X3> on the layer
shapes of HTD-R50 @ B = 4, 800x1344: forward and data-gradient TFLOP/s per tile configuration of the new kernel
(HTD_X3P_FORCE_TILE; HTD_X3P_TUNE=1 must be set at load time), the old kernel's rate beside it, and the largest difference
between the two results (same arithmetic, different summation order)."""
import os
import sys

os.environ['HTD_X3P_TUNE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import capi, dense

CL = torch.channels_last
_P, _S = capi.ptr, capi.current_stream_ptr
LAYERS = [  # name, Ci, H, W, Co, k, stride, pad
    ('fpn P2 3x3 256', 256, 200, 336, 256, 3, 1, 1),
    ('fpn P3 3x3 256', 256, 100, 168, 256, 3, 1, 1),
    ('fpn P4 3x3 256', 256, 50, 84, 256, 3, 1, 1),
    ('l1.conv2 3x3 64', 64, 200, 336, 64, 3, 1, 1),
    ('l2.conv2 3x3 128', 128, 100, 168, 128, 3, 1, 1),
    ('l3.conv2 3x3 256', 256, 50, 84, 256, 3, 1, 1),
    ('l4.conv2 3x3 512', 512, 25, 42, 512, 3, 1, 1),
    ('l1.conv3 1x1 64-256', 64, 200, 336, 256, 1, 1, 0),
    ('l1.conv1 1x1 256-64', 256, 200, 336, 64, 1, 1, 0),
    ('l2.conv1 1x1 512-128', 512, 100, 168, 128, 1, 1, 0),
    ('l2.conv3 1x1 128-512', 128, 100, 168, 512, 1, 1, 0),
    ('l3.conv1 1x1 1024-256', 1024, 50, 84, 256, 1, 1, 0),
    ('l3.conv3 1x1 256-1024', 256, 50, 84, 1024, 1, 1, 0),
    ('l4.conv1 1x1 2048-512', 2048, 25, 42, 512, 1, 1, 0),
    ('l4.conv3 1x1 512-2048', 512, 25, 42, 2048, 1, 1, 0),
    ('fpn lat 1x1 2048-256', 2048, 25, 42, 256, 1, 1, 0),
    ('fpn lat 1x1 512-256', 512, 100, 168, 256, 1, 1, 0),
    ('l3 ds 1x1 512-1024 s2', 512, 100, 168, 1024, 1, 2, 0),
    ('reg conv 3x3 576 n=512', 576, 7 * 16, 7 * 32, 576, 3, 1, 1),
    ('reg conv 576 n=28 (7x7)', 576, 7, 7, 576, 3, 1, 1),
    ('fc 12544-1024 M=2048', 12544, 32, 16, 1024, 1, 1, 0),
    ('fc 1024-1024 M=2048', 1024, 32, 16, 1024, 1, 1, 0),
    ('fc dgrad 1024-12544', 1024, 32, 16, 12544, 1, 1, 0),
]
NAMES = ['auto', '64x64', '128x128', '128x64', '64x128']
SHAPES = [s for s in os.environ.get('SWEEP_MFMA', '0').split(',')]          # 0: the launcher's choice per filter width
NBS = [s for s in os.environ.get('SWEEP_NB', '0').split(',')]               # B buffers (-DHTD_X3P_DEEP builds); 0: default
REMS = [s for s in os.environ.get('SWEEP_REM', '1,0').split(',')]           # K splits of the remainder tiles: 1 none, 0 planned, n forced


def timed(fn, key, flop, n=6):
    for it in range(n + 2):
        if it == 2:
            capi.profile_begin()
        fn()
    calls, ms = capi.profile_end()[key][:2]
    return flop / (ms / calls * 1e-3) / 1e12


def old_fwd(x, w, s, p):
    B, Ci, H, W = x.shape
    Co, _, k, _ = w.shape
    Ho, Wo = dense._out_hw(H, W, k, k, s, p, 1)
    y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=CL)
    capi.call('htd_conv2d_fwd', _P(x), _P(w), None, None, 0, 0, _P(y), B, H, W, Ci, Co, k, k, s, p, 1, 0,
              _P(dense._splitk_ws(B * Ho * Wo, Co, Ci, k, k, x.device)), _S())
    return y


def main():
    dev = torch.device('cuda:0')
    only = sys.argv[1] if len(sys.argv) > 1 else None
    B = int(os.environ.get('SWEEP_B', '4'))
    print(f'{"layer":26s} {"GFLOP":>7s} | {"igemm":>7s} | ' + ' '.join(f'{n:>8s}' for n in NAMES) + ' | fwd: new TF/s per tile, max |new - old|')
    with torch.no_grad():
        for name, Ci, H, W, Co, k, s, p in LAYERS:
            if only and only not in name:
                continue
            Bn = 28 if 'n=28' in name else (1 if name.startswith('reg conv 3x3') else B)
            x = torch.randn(Bn, Ci, H, W, device=dev).contiguous(memory_format=CL)
            w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5).contiguous(memory_format=CL)
            os.environ['HTD_X3P_FORCE_TILE'] = '-1'
            y_old = old_fwd(x, w, s, p)
            flop = 2.0 * y_old.numel() * Ci * k * k
            r_old = timed(lambda: old_fwd(x, w, s, p), 'htd_conv2d_fwd', flop)
            for shape, nb, rem in [(a, b, c) for a in SHAPES for b in NBS for c in REMS]:
                os.environ['HTD_X3P_MFMA'] = shape
                os.environ['HTD_X3P_NB'] = nb
                os.environ['HTD_X3P_REM_SPLITS'] = rem
                out, err = [], 0.0
                for cfg in range(-1, 4):
                    os.environ['HTD_X3P_FORCE_TILE'] = str(cfg)
                    y = dense._fwd_raw(x, w, None, None, s, p, 1, False)
                    err = max(err, float((y - y_old).abs().max()))
                    out.append(timed(lambda: dense._fwd_raw(x, w, None, None, s, p, 1, False), 'htd_conv2d_fwd_x3p', flop))
                os.environ['HTD_X3P_FORCE_TILE'] = '-1'
                line = f'{name:22s} m{shape:2s} nb{nb} rem{rem:2s} {flop / 1e9:7.1f} | {r_old:7.1f} | ' + ' '.join(f'{r:8.1f}' for r in out) + \
                    f' | best/old {max(out[1:]) / r_old:.2f} auto/best {out[0] / max(out[1:]):.2f} err {err:.2e}'
                if s == 1:       # data gradient of the same layer (new kernel, automatic tile)
                    g = torch.randn_like(y_old)
                    r_dg = timed(lambda: dense._dgrad_raw(g, w, x.shape, 1, p, 1), 'htd_conv2d_bwd_data_x3p', flop)
                    line += f' | dgrad {r_dg:6.1f}'
                print(line, flush=True)
            os.environ['HTD_X3P_MFMA'] = os.environ['HTD_X3P_NB'] = os.environ['HTD_X3P_REM_SPLITS'] = '0'


if __name__ == '__main__':
    main()
