#!/usr/bin/env python3
"""Random-shape check of the forward / data-gradient convolution kernels and their epilogues (bias, residual, ReLU, the ReLU
mask of the backward) on small-integer operands, where every product and sum is exactly representable: the result of
dense.conv2d and of its backward must EQUAL the fp64 reference.  Covers conv_x3p_kernel (1x1 any stride, 3x3 stride 1),
conv_igemm_kernel (the other shapes, Co < 33) and the stem.

    python tools/fuzz_conv.py [n_shapes] [seed]
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from htd_amd import dense

CL = torch.channels_last


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    dev = torch.device('cuda', 0)
    bad = done = 0
    while done < n:
        k = rng.choice([1, 1, 3, 3, 3])
        s = rng.choice([1, 1, 1, 2])
        p = k // 2
        Ci = rng.choice([16, 32, 48, 64, 128, 256, 8, 24])
        Co = rng.choice([15, 16, 36, 64, 81, 128, 132, 256, 512])
        B, H, W = rng.randint(1, 4), rng.randint(1, 40), rng.randint(1, 60)
        if rng.random() < 0.25:
            k, s, p, H, W, B = 1, 1, 0, 1, 1, rng.randint(1, 2500)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        if Ho <= 0 or Wo <= 0:
            continue
        done += 1
        g = torch.Generator().manual_seed(done * 31 + seed)
        relu, with_res, with_bias = rng.random() < 0.6, rng.random() < 0.5, rng.random() < 0.7
        x = torch.randint(-3, 4, (B, Ci, H, W), generator=g).float().to(dev).contiguous(memory_format=CL).requires_grad_()
        w = torch.randint(-2, 3, (Co, Ci, k, k), generator=g).float().to(dev).contiguous(memory_format=CL).requires_grad_()
        b = torch.randint(-3, 4, (Co, ), generator=g).float().to(dev).requires_grad_() if with_bias else None
        r = torch.randint(-5, 6, (B, Co, Ho, Wo), generator=g).float().to(dev).contiguous(memory_format=CL).requires_grad_() if with_res else None
        gy = torch.randint(-3, 4, (B, Co, Ho, Wo), generator=g).float().to(dev).contiguous(memory_format=CL)
        dense.new_step()
        y = dense.conv2d(x, w, b, s, p, 1, relu=relu, residual=r)
        y.backward(gy)
        xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
        bd = b.detach().double().requires_grad_() if with_bias else None
        rd = r.detach().double().requires_grad_() if with_res else None
        ref = F.conv2d(xd, wd, bd, s, p)
        if with_res:
            ref = ref + rd
        if relu:
            ref = F.relu(ref)
        ref.backward(gy.double())
        ok = torch.equal(y.detach().double(), ref.detach()) and torch.equal(x.grad.double(), xd.grad) and \
            torch.equal(w.grad.double(), wd.grad)
        if with_bias:
            ok = ok and torch.equal(b.grad.double(), bd.grad)
        if with_res:
            ok = ok and torch.equal(r.grad.double(), rd.grad)
        if not ok:
            bad += 1
            print('MISMATCH', (B, Ci, H, W, Co, k, s, p), 'relu', relu, 'res', with_res, 'bias', with_bias,
                  float((y.detach().double() - ref.detach()).abs().max()), float((x.grad.double() - xd.grad).abs().max()),
                  float((w.grad.double() - wd.grad).abs().max()))
    print(f'{n} convolution problems checked (forward, data gradient, weight gradient, epilogues), {bad} mismatches')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
