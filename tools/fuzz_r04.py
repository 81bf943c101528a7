#!/usr/bin/env python3
"""Random-shape checks of the round-4 kernels on small-integer operands (every product and sum exactly representable, so results
must EQUAL the fp64 reference whatever the summation order):
  planes   3x3 layer writing activation planes -> plane-fed 1x1 layer (conv_x3q_kernel) with bias / residual / ReLU, and the
           mirrored pair of the backward (3x3 data gradient writing planes -> plane-fed 1x1 data gradient with mask / accum)
  bf16q    conv_bf16q_kernel: 1x1 (any stride) and 3x3 / stride 1, forward and data-gradient form
  wacc     htd_conv2d_bwd_weight_acc: a weight gradient added into a slice that already holds another one
  heads    htd_rpn_heads_gather / _scatter against permute / reshape / cat
usage: python tools/fuzz_r04.py [n_per_kind] [seed]"""
import ctypes
import os
import random
import sys

os.environ.setdefault('HTD_BF16Q', '2')          # conv_bf16q_kernel takes every shape it can
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from htd_amd import capi, dense

CL = torch.channels_last
BF = torch.bfloat16
P, S = capi.ptr, capi.current_stream_ptr


def ints(g, shape, lo, hi, dev):
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = random.Random(seed)
    dev = torch.device('cuda', 0)
    L = capi.lib()
    bad = {'planes': 0, 'bf16q': 0, 'wacc': 0, 'heads': 0}
    for it in range(n):
        g = torch.Generator().manual_seed(1000 * seed + it)
        # ---- planes
        Cm, Co = rng.choice([16, 32, 48, 64, 128, 256]), rng.choice([33, 48, 64, 96, 160, 256, 512, 1024])
        B, H, W = rng.randint(1, 4), rng.randint(1, 30), rng.randint(1, 44)
        if L.htd_conv2d_x3p_supported(Cm, Cm, 3, 3, 1, 1, 1) and L.htd_conv2d_x3p_supported(Cm, Co, 1, 1, 1, 0, 1):
            x = ints(g, (B, Cm, H, W), -2, 2, dev).contiguous(memory_format=CL)
            w2 = ints(g, (Cm, Cm, 3, 3), -1, 1, dev).contiguous(memory_format=CL)
            w3 = ints(g, (Co, Cm, 1, 1), -2, 2, dev).contiguous(memory_format=CL)
            b2, b3 = ints(g, (Cm, ), -3, 3, dev), ints(g, (Co, ), -3, 3, dev)
            res = ints(g, (B, Co, H, W), -5, 5, dev).contiguous(memory_format=CL)
            dense.new_step()
            h, hp = dense._fwd_raw(x, w2, b2, None, 1, 1, 1, True, emit=True)
            y = dense._fwd_raw(h, w3, b3, res, 1, 0, 1, True, x_planes=hp)
            href = F.relu(F.conv2d(x.double(), w2.double(), b2.double(), 1, 1))
            yref = F.relu(F.conv2d(href, w3.double(), b3.double()) + res.double())
            ok = torch.equal(h.double(), href) and torch.equal(y.double(), yref)
            if L.htd_conv2d_x3p_supported(Cm, Co, 1, 1, 1, 0, 1):
                w1 = ints(g, (Cm, Co, 1, 1), -1, 1, dev).contiguous(memory_format=CL)
                gy = ints(g, (B, Cm, H, W), -2, 2, dev).contiguous(memory_format=CL)
                msk, acc = ints(g, (B, Co, H, W), -1, 1, dev).contiguous(memory_format=CL), ints(g, (B, Co, H, W), -3, 3, dev).contiguous(memory_format=CL)
                gm, gmp = dense._dgrad_raw(gy, w2, (B, Cm, H, W), 1, 1, 1, mask_src=x, emit=True)
                gx = dense._dgrad_raw(gm, w1, (B, Co, H, W), 1, 0, 1, mask_src=msk, accum=acc, g_planes=gmp)
                gmref = torch.nn.grad.conv2d_input(x.shape, w2.double(), gy.double(), 1, 1) * (x.double() > 0)
                gxref = (torch.nn.grad.conv2d_input((B, Co, H, W), w1.double(), gmref, 1, 0) + acc.double()) * (msk.double() > 0)
                ok = ok and torch.equal(gm.double(), gmref) and torch.equal(gx.double(), gxref)
            if not ok:
                bad['planes'] += 1
                print('MISMATCH planes', (B, Cm, H, W, Co))
        # ---- bf16q
        k = rng.choice([1, 1, 3])
        s = rng.choice([1, 1, 2, 3]) if k == 1 else 1
        Ci, Cq = rng.choice([64, 128, 192, 256, 576]), rng.choice([4, 36, 64, 96, 128, 260, 512])
        B, H, W = rng.randint(1, 4), rng.randint(1, 30), rng.randint(1, 44)
        os.environ['HTD_BF16Q_TILE'] = rng.choice(['64', '128'])
        xi, wi = ints(g, (B, Ci, H, W), -2, 2, dev), ints(g, (Cq, Ci, k, k), -1, 1, dev)
        y = dense.conv2d_bf16(xi.to(BF).contiguous(memory_format=CL), wi.to(BF).contiguous(memory_format=CL), None, s, k // 2, 1)
        if not torch.equal(y.float(), F.conv2d(xi, wi, None, s, k // 2).to(BF).float()):
            bad['bf16q'] += 1
            print('MISMATCH bf16q', (B, Ci, H, W, Cq, k, s))
        # ---- accumulating weight gradient
        k = rng.choice([1, 3])
        Ci, Cw = rng.choice([16, 64, 128, 256]), rng.choice([16, 64, 128, 256])
        outs = []
        gw = torch.zeros(Cw, Ci, k, k, device=dev).contiguous(memory_format=CL)
        gb = torch.zeros(Cw, device=dev)
        ref_w, ref_b = torch.zeros(Cw, Ci, k, k, dtype=torch.float64, device=dev), torch.zeros(Cw, dtype=torch.float64, device=dev)
        for call in range(3):
            B, H, W = rng.randint(1, 3), rng.randint(1, 24), rng.randint(1, 30)
            x = ints(g, (B, Ci, H, W), -2, 2, dev).contiguous(memory_format=CL)
            gy = ints(g, (B, Cw, H, W), -2, 2, dev).contiguous(memory_format=CL)
            nb = L.htd_conv2d_wgrad_workspace_bytes(B, H, W, Ci, Cw, k, k, 1, k // 2, 1)
            ws = torch.empty(nb // 4 + 1, device=dev)
            capi.call('htd_conv2d_bwd_weight_acc' if call else 'htd_conv2d_bwd_weight', P(x), P(gy), P(gw), P(gb), B, H, W, Ci, Cw, k, k,
                      1, k // 2, 1, P(ws), S())
            ref_w += torch.nn.grad.conv2d_weight(x.double(), (Cw, Ci, k, k), gy.double(), 1, k // 2)
            ref_b += gy.double().sum((0, 2, 3))
        if not (torch.equal(gw.double(), ref_w) and torch.equal(gb.double(), ref_b)):
            bad['wacc'] += 1
            print('MISMATCH wacc', (Ci, Cw, k))
        # ---- RPN head gather / scatter
        Lv, na, Bq = rng.randint(1, 5), rng.choice([1, 3]), rng.randint(1, 3)
        C = 5 * na + rng.choice([0, 1, 3])
        sizes = [(rng.randint(1, 20), rng.randint(1, 30)) for _ in range(Lv)]
        ys = [torch.randn(Bq, C, h, w, generator=g).to(dev).contiguous(memory_format=CL) for h, w in sizes]
        pix = [h * w for h, w in sizes]
        A = na * sum(pix)
        cls, reg = torch.empty(Bq, A, device=dev), torch.empty(Bq, A, 4, device=dev)
        capi.call('htd_rpn_heads_gather', (ctypes.c_void_p * Lv)(*[t.data_ptr() for t in ys]), (ctypes.c_int64 * Lv)(*pix), Lv, Bq, C, na,
                  P(cls), P(reg), S())
        rc = torch.cat([t[:, :na].permute(0, 2, 3, 1).reshape(Bq, -1) for t in ys], 1)
        rr = torch.cat([t[:, na:5 * na].permute(0, 2, 3, 1).reshape(Bq, -1, 4) for t in ys], 1)
        gys = [torch.full_like(t, 7.0) for t in ys]
        capi.call('htd_rpn_heads_scatter', P(cls), P(reg), (ctypes.c_void_p * Lv)(*[t.data_ptr() for t in gys]), (ctypes.c_int64 * Lv)(*pix),
                  Lv, Bq, C, na, S())
        back = all(torch.equal(gt[:, :5 * na], t[:, :5 * na]) and bool((gt[:, 5 * na:] == 0).all()) for gt, t in zip(gys, ys))
        if not (torch.equal(cls, rc) and torch.equal(reg, rr) and back):
            bad['heads'] += 1
            print('MISMATCH heads', sizes, na, C)
    print(f'{n} problems per kind checked:', ', '.join(f'{k} {v} mismatches' for k, v in bad.items()))
    sys.exit(1 if any(bad.values()) else 0)


if __name__ == '__main__':
    main()
