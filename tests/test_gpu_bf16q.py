"""conv_bf16q_kernel (csrc/conv_bf16.hip, round 4): bf16 convolution with both operands by LDS-DMA into swizzled row images and a
ring of tiles in flight -- the kernel under htd_conv2d_fwd_bf16 / _fwd_bf16_up / _dgrad_bf16 for 1x1 layers (any stride) and
3x3 / stride-1 layers with Ci % 64 == 0.  On small-integer operands every product and partial sum is exact, so ANY wrong chunk
(the source-side XOR swizzle), halo row, border mask, ring buffer or tap shows as a bit difference against the fp32 convolution;
on real data the two bf16 kernels differ only in summation order.  Role in the reference: cuDNN's half-precision convolutions
under mmcv's fp16 hook (mmdet/core/fp16/, backbones/resnet.py:260-300)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, %r)
from htd_amd import capi, dense
CL = torch.channels_last
BF = torch.bfloat16
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(2)
P, S = capi.ptr, capi.current_stream_ptr
bad = []
CASES = [  # B, Ci, H, W, Co, k, stride: borders, ragged tiles in M and Co, narrow maps whose runs wrap rows / images, strides
    (2, 64, 20, 28, 128, 3, 1), (1, 256, 13, 17, 256, 3, 1), (5, 576, 7, 7, 576, 3, 1), (3, 64, 7, 7, 96, 3, 1),
    (1, 64, 1, 1, 64, 3, 1), (1, 64, 3, 200, 40, 3, 1), (1, 128, 130, 130, 256, 3, 1), (2, 64, 5, 3, 36, 3, 1),
    (2, 64, 20, 28, 64, 1, 1), (2, 256, 20, 28, 512, 1, 2), (3, 64, 9, 11, 576, 1, 3), (1, 1024, 37, 5, 256, 1, 1),
    (4, 256, 50, 84, 1024, 1, 1), (1, 12544, 37, 1, 1024, 1, 1),
]
for B, Ci, H, W, Co, k, s in CASES:
    p = k // 2
    xi = torch.randint(-2, 3, (B, Ci, H, W), generator=g).float()
    wi = torch.randint(-1, 2, (Co, Ci, k, k), generator=g).float()
    ref = F.conv2d(xi, wi, None, s, p).to(BF)
    x = xi.to(dev).to(BF).contiguous(memory_format=CL)
    w = wi.to(dev).to(BF).contiguous(memory_format=CL)
    xr = torch.randn(B, Ci, H, W, generator=g).to(dev).to(BF).contiguous(memory_format=CL)
    wr = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).to(dev).to(BF).contiguous(memory_format=CL)
    bias = torch.randn(Co, generator=g).to(dev)
    res = torch.randn(tuple(ref.shape), generator=g).to(dev).to(BF).contiguous(memory_format=CL)
    os.environ['HTD_BF16Q'] = '0'
    old = dense.conv2d_bf16(xr, wr, bias, s, p, 1, True, res)
    yref = (F.conv2d(xr.float(), wr.float(), bias, s, p) + res.float()).relu()
    os.environ['HTD_BF16Q'] = '1'
    for tile in (64, 128):
        for ns in (2, 3, 4):
            if ns == 4 and (k == 3 or tile == 128):
                continue
            os.environ['HTD_BF16Q_TILE'], os.environ['HTD_BF16Q_NS'] = str(tile), str(ns)
            y = dense.conv2d_bf16(x, w, None, s, p, 1)
            if not torch.equal(y.cpu().float(), ref.float()):
                bad.append(('int', B, Ci, H, W, Co, k, s, tile, ns, int((y.cpu().float() != ref.float()).sum())))
            y = dense.conv2d_bf16(xr, wr, bias, s, p, 1, True, res)
            # same products, fp32 accumulation in another order, one rounding to bf16: within a bf16 ulp of the fp32 result
            err = float(((y.float() - yref).abs() / (yref.abs() + 1.0)).max())
            if err > 2 ** -7 or float((y.float() - old.float()).abs().max()) > 2 ** -6 * float(yref.abs().max()):
                bad.append(('real', B, Ci, H, W, Co, k, s, tile, ns, err))
            if s == 1 and Co %% 64 == 0:
                # data-gradient form: transposed / flipped weights, mask and accum in the epilogue
                gy = torch.randint(-2, 3, tuple(ref.shape), generator=g).float()
                gref = torch.nn.grad.conv2d_input(xi.shape, wi, gy, 1, p)
                wb, wT = dense._prep_bf16(wi.to(dev).contiguous(memory_format=CL))
                gx = dense._dgrad_bf16_raw(gy.to(dev).to(BF).contiguous(memory_format=CL), wT, k, p, 1)
                if not torch.equal(gx.cpu().float(), gref.to(BF).float()):
                    bad.append(('dgrad', B, Ci, H, W, Co, k, s, tile, ns))
                msk = torch.randint(-1, 2, xi.shape, generator=g).float().to(dev).to(BF).contiguous(memory_format=CL)
                acc = torch.randint(-2, 3, xi.shape, generator=g).float().to(dev).to(BF).contiguous(memory_format=CL)
                gx = dense._dgrad_bf16_raw(gy.to(dev).to(BF).contiguous(memory_format=CL), wT, k, p, 1, mask_src=msk, accum=acc)
                want = ((gref.to(dev) + acc.float()) * (msk.float() > 0)).to(BF)
                if not torch.equal(gx.float(), want.float()):
                    bad.append(('dgrad-epi', B, Ci, H, W, Co, k, s, tile, ns))
print('BAD', bad[:20], len(bad))
sys.exit(1 if bad else 0)
'''


def test_lds_dma_bf16_kernel_is_exact_on_integers_for_every_tile_and_ring_depth():
    env = dict(os.environ, HTD_BF16Q_TUNE='1')
    r = subprocess.run([sys.executable, '-c', _SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
