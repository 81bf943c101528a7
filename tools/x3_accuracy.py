#!/usr/bin/env python3
"""Error of the three product arithmetics of the fp32 convolutions against an fp64 reference, on wide-dynamic-range data
(the inputs of tests/test_gpu_conv.py::test_split_bf16_products_are_fp32_accurate), over several seeds:
   native  v_mfma_f32_32x32x2_f32 (htd_conv2d_set_math(0))
   x3      conv_igemm_kernel<X3>: split per tile and tap (HTD_X3P=0)
   x3p     conv_x3p_kernel: weights pre-split, halo runs (csrc/conv_x3.hip), six bf16 products
   x3h     the same kernel on the H2 arithmetic (3x3 layers): two block-scaled fp16 pieces per operand, three products
Printed per case: max and rms of |result - fp64| / accumulated magnitude, forward and data gradient."""
import os
import sys

os.environ['HTD_X3P_TUNE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import capi, dense

CASES = [(256, 256, 3, 40, 56), (1024, 256, 1, 50, 84), (64, 256, 1, 60, 80), (48, 96, 3, 33, 47), (576, 576, 3, 7, 7)]
if os.environ.get('X3_ACC_CASES') == '3x3':
    CASES = [c for c in CASES if c[2] == 3]
SEEDS = [0, 1, 2, 3, 4]
MODES = ('native', 'x3', 'x3p', 'x3h')


def main():
    dev = torch.device('cuda:0')
    L = capi.lib()
    for Ci, Co, k, H, W in CASES:
        rows = {m: [] for m in MODES}
        for seed in SEEDS:
            g = torch.Generator().manual_seed(Ci + k + 1000 * seed)
            B = 2 if H > 7 else 16
            x = (torch.randn(B, Ci, H, W, generator=g) * torch.exp(torch.randn(B, Ci, H, W, generator=g) * 2)).to(dev)
            w = (torch.randn(Co, Ci, k, k, generator=g) * torch.exp(torch.randn(Co, Ci, k, k, generator=g))).to(dev) / (Ci * k * k) ** 0.5
            x = x.contiguous(memory_format=torch.channels_last)
            w = w.contiguous(memory_format=torch.channels_last)
            p = k // 2
            ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, p)
            scale = torch.nn.functional.conv2d(x.double().abs(), w.double().abs(), None, 1, p)
            gy = torch.randn(ref.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
            gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 1, p)
            gscale = torch.nn.grad.conv2d_input(x.shape, w.double().abs(), gy.double().abs(), 1, p)
            for mode in MODES:
                L.htd_conv2d_set_math(0 if mode == 'native' else 1)
                L.htd_conv2d_set_h2(1 if mode == 'x3h' else 0)
                os.environ['HTD_X3P'] = '1' if mode in ('x3p', 'x3h') else '0'
                dense.new_step()
                xr = x.clone().requires_grad_()
                gyr = gy.clone()
                if mode == 'x3h':             # as tensors written by the package's epilogues: both carry their maxima
                    dense.tag_amax(xr, dense.absmax(xr))
                    dense.tag_amax(gyr, dense.absmax(gyr))
                y = dense.conv2d(xr, w, None, 1, p, 1)
                y.backward(gyr)
                ef = (y.detach().double() - ref).abs() / scale
                eg = (xr.grad.double() - gref).abs() / gscale
                rows[mode].append((float(ef.max()), float(ef.pow(2).mean().sqrt()), float(eg.max()), float(eg.pow(2).mean().sqrt())))
            L.htd_conv2d_set_math(1)
            L.htd_conv2d_set_h2(1)
        print(f'case Ci={Ci} Co={Co} k={k} {H}x{W}  (K fwd {Ci * k * k}, dgrad {Co * k * k})')
        for mode, r in rows.items():
            if not r:
                continue
            t = torch.tensor(r)
            print(f'  {mode:7s} fwd max {t[:, 0].mean():.2e} (worst seed {t[:, 0].max():.2e}) rms {t[:, 1].mean():.2e} | '
                  f'dgrad max {t[:, 2].mean():.2e} (worst {t[:, 2].max():.2e}) rms {t[:, 3].mean():.2e}')
        tn, tp = torch.tensor(rows['native']), torch.tensor(rows['x3p'])
        print('  x3p / native per seed: fwd max', [round(float(a / b), 2) for a, b in zip(tp[:, 0], tn[:, 0])],
              'dgrad max', [round(float(a / b), 2) for a, b in zip(tp[:, 2], tn[:, 2])],
              'rms', [round(float(a / b), 2) for a, b in zip(tp[:, 3], tn[:, 3])], flush=True)
        if rows['x3h']:
            th = torch.tensor(rows['x3h'])
            print('  x3h / native per seed: fwd max', [round(float(a / b), 2) for a, b in zip(th[:, 0], tn[:, 0])],
                  'fwd rms', [round(float(a / b), 2) for a, b in zip(th[:, 1], tn[:, 1])],
                  'dgrad max', [round(float(a / b), 2) for a, b in zip(th[:, 2], tn[:, 2])],
                  'rms', [round(float(a / b), 2) for a, b in zip(th[:, 3], tn[:, 3])], flush=True)


if __name__ == '__main__':
    main()
