"""fp32 MFMA implicit-GEMM convolution / Linear kernels against a plain PyTorch fp64 CPU reference.
Tolerance: fp32 accumulation over K terms, relative 2e-5 * sqrt(K)-ish => rtol 1e-4, atol scaled."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last

CASES = [
    # B, Ci, H, W, Co, k, stride, pad, dil
    (2, 64, 20, 28, 64, 1, 1, 0, 1),
    (2, 64, 20, 28, 128, 3, 1, 1, 1),
    (1, 256, 13, 17, 256, 3, 1, 1, 1),
    (2, 128, 21, 30, 128, 3, 2, 1, 1),
    (2, 256, 20, 28, 512, 1, 2, 0, 1),
    (2, 3, 32, 40, 64, 7, 2, 3, 1),
    (1, 256, 25, 31, 15, 1, 1, 0, 1),
    (3, 576, 7, 7, 576, 3, 1, 1, 1),
    (1, 16, 9, 9, 40, 3, 1, 2, 2),
    (1, 24, 5, 6, 8, 3, 1, 1, 1),
    # 265 / 300 / 1058 tiles: a few tiles more than a multiple of 256 -> the balanced tail (whole tiles + K-split remainder)
    (1, 32, 130, 130, 64, 3, 1, 1, 1),
    (2, 64, 98, 98, 64, 1, 1, 0, 1),
    (1, 64, 130, 130, 256, 1, 1, 0, 1),
    (1, 32, 130, 130, 256, 3, 1, 1, 1),
]


@pytest.mark.parametrize('B,Ci,H,W,Co,k,stride,pad,dil', CASES)
@pytest.mark.parametrize('relu,with_res', [(False, False), (True, True)])
def test_conv2d_fwd_bwd(B, Ci, H, W, Co, k, stride, pad, dil, relu, with_res):
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B * 1000 + Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ref = F.conv2d(xr, wr, br, stride, pad, dil)
    res = torch.randn(ref.shape, generator=g) if with_res else None
    rr = res.double().requires_grad_() if with_res else None
    if with_res:
        ref = ref + rr
    if relu:
        ref = F.relu(ref)
    xd = x.to(dev).contiguous(memory_format=CL).requires_grad_()
    wd = w.to(dev).contiguous(memory_format=CL).requires_grad_()
    bd = b.to(dev).requires_grad_()
    rd = res.to(dev).contiguous(memory_format=CL).requires_grad_() if with_res else None
    y = dense.conv2d(xd, wd, bd, stride, pad, dil, relu, rd)
    assert y.shape == ref.shape
    K = Ci * k * k
    torch.testing.assert_close(y.cpu().double(), ref.detach(), rtol=1e-4, atol=1e-5 * K ** 0.5)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go.double())
    y.backward(go.to(dev))
    npix = ref.shape[0] * ref.shape[2] * ref.shape[3]
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-4, atol=2e-5 * (Co * k * k) ** 0.5)
    torch.testing.assert_close(wd.grad.cpu().double(), wr.grad, rtol=1e-4, atol=2e-5 * npix ** 0.5)
    torch.testing.assert_close(bd.grad.cpu().double(), br.grad, rtol=1e-4, atol=2e-5 * npix ** 0.5)
    if with_res:
        torch.testing.assert_close(rd.grad.cpu().double(), rr.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('M,K,N', [(37, 12544, 1024), (200, 1024, 81), (5, 256, 128), (64, 81, 1025), (1, 128, 1)])
def test_linear(M, K, N):
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(M + K + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ref = F.relu(F.linear(xr, wr, br))
    xd, wd, bd = x.to(dev).requires_grad_(), w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    y = dense.linear(xd, wd, bd, relu=True)
    torch.testing.assert_close(y.cpu().double(), ref.detach(), rtol=1e-4, atol=1e-5 * K ** 0.5)
    go = torch.randn(M, N, generator=g)
    ref.backward(go.double())
    y.backward(go.to(dev))
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-4, atol=2e-5 * N ** 0.5)
    torch.testing.assert_close(wd.grad.cpu().double(), wr.grad, rtol=1e-4, atol=2e-5 * M ** 0.5)
    torch.testing.assert_close(bd.grad.cpu().double(), br.grad, rtol=1e-4, atol=2e-5 * M ** 0.5)


def test_conv_rejects_cpu():
    from htd_amd import dense
    with pytest.raises(NotImplementedError):
        dense.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 1, 1))


@pytest.mark.gpu
@pytest.mark.parametrize('stride,dilation,needs_x', [(2, 1, True), (1, 1, False), (1, 2, True)])
def test_res_stage_fused_backward(stride, dilation, needs_x):
    """ResLayer as one autograd node (dense.ResStageFunction: ReLU masks and the residual join inside the dgrad
    epilogues) against the same layer run conv by conv through autograd, and against ATen in fp64-free fp32."""
    import torch.nn as nn
    from htd_amd.detector.resnet import Bottleneck, ResLayer
    torch.manual_seed(3)
    dev = torch.device('cuda:0')
    layer = ResLayer(Bottleneck, 64, 32, 3, stride=stride, dilation=dilation).to(dev)
    for m in layer.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    layer.eval()                                                     # frozen statistics, parameters still trainable
    x = torch.randn(2, 64, 20, 28, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(needs_x)
    gy = None
    results = []
    for fused in (True, False):
        layer.zero_grad()
        if x.grad is not None:
            x.grad = None
        y = layer(x) if fused else nn.Sequential.forward(layer, x)
        if gy is None:
            gy = torch.randn_like(y)
        y.backward(gy)
        results.append((y.detach().clone(), None if not needs_x else x.grad.clone(),
                        {n: p.grad.clone() for n, p in layer.named_parameters()}))
    (y0, gx0, gp0), (y1, gx1, gp1) = results
    assert torch.equal(y0, y1)
    if needs_x:
        torch.testing.assert_close(gx0, gx1, rtol=1e-5, atol=1e-6)
    assert set(gp0) == set(gp1) and len(gp0) > 0
    for n in gp0:
        torch.testing.assert_close(gp0[n], gp1[n], rtol=1e-5, atol=1e-5 * float(gp1[n].abs().max()), msg=n)


@pytest.mark.gpu
def test_fpn_fused_top_down_matches_interpolate_add():
    """FPN with the top-down `laterals[i-1] += interpolate(laterals[i])` folded into the lateral conv epilogues
    (residual read through nearest up-sampling, incl. an odd size 25 -> 13) against the unfused module."""
    from htd_amd.detector.fpn import FPN
    torch.manual_seed(4)
    dev = torch.device('cuda:0')
    fpn = FPN([32, 64, 128, 256], 64, 5).to(dev)
    fpn.init_weights()
    sizes = [(50, 84), (25, 42), (13, 21), (7, 11)]
    xs = [torch.randn(2, c, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
          for c, (h, w) in zip([32, 64, 128, 256], sizes)]
    res = []
    for fused in (True, False):
        fpn.fused_top_down = fused
        fpn.zero_grad()
        for x in xs:
            x.grad = None
        outs = fpn(xs)
        loss = sum((o * torch.linspace(0.5, 1.5, o.numel(), device=dev).view_as(o)).sum() for o in outs)
        loss.backward()
        res.append(([o.detach().clone() for o in outs], [x.grad.clone() for x in xs],
                    {n: p.grad.clone() for n, p in fpn.named_parameters()}))
    (o1, g1, p1), (o2, g2, p2) = res
    # (the two formulations need not run a layer on the same product arithmetic -- H2 where the input carries its maximum: the fused
    # laterals' outputs do, the sum torch makes in the unfused module does not and is too small for a pass of its own; the
    # three-piece bf16 form elsewhere: both fp32-accurate, not bit-equal)
    for a, b in zip(o1, o2):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=4e-5)
    for n in p2:
        torch.testing.assert_close(p1[n], p2[n], rtol=1e-4, atol=1e-4 * float(p2[n].abs().max()), msg=n)


@pytest.mark.gpu
def test_bf16_fpn_top_down_runs_inside_the_lateral_convolutions():
    """bf16 maps: the up-sampled residual of htd_conv2d_fwd_bf16_up and htd_upsample_nearest_bwd_bf16 (odd sizes 25 -> 13 ->
    7) are EXACT on small integers (every sum is representable), and the fused FPN agrees with interpolate + add to the one
    bf16 rounding the fused form saves."""
    from htd_amd import capi, dense
    from htd_amd.detector.fpn import FPN
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    CL = torch.channels_last
    for (H, W), (h, w) in (((25, 42), (13, 21)), ((13, 21), (7, 11)), ((50, 84), (25, 42))):
        x = torch.randint(-2, 3, (2, 64, H, W), generator=g).to(dev, torch.bfloat16).contiguous(memory_format=CL)
        wt = torch.randint(-1, 2, (72, 64, 1, 1), generator=g).to(dev, torch.bfloat16).contiguous(memory_format=CL)
        b = torch.randint(-3, 4, (72, ), generator=g).float().to(dev)
        r = torch.randint(-9, 10, (2, 72, h, w), generator=g).to(dev, torch.bfloat16).contiguous(memory_format=CL)
        y = dense.conv2d_bf16(x, wt, b, 1, 0, 1, False, r, res_up=True)
        ref = F.conv2d(x.float(), wt.float(), b) + F.interpolate(r.float(), size=(H, W), mode='nearest')
        assert torch.equal(y.float(), ref)
        gy = torch.randint(-4, 5, (2, 72, H, W), generator=g).to(dev, torch.bfloat16).contiguous(memory_format=CL)
        gr = torch.empty((2, 72, h, w), device=dev, dtype=torch.bfloat16).contiguous(memory_format=CL)
        capi.call('htd_upsample_nearest_bwd_bf16', dense._P(gy), dense._P(gr), 2, H, W, h, w, 72, dense._S())
        rf = r.float().requires_grad_()
        F.interpolate(rf, size=(H, W), mode='nearest').backward(gy.float())
        assert torch.equal(gr.float(), rf.grad)
    torch.manual_seed(4)
    fpn = FPN([32, 64, 128, 256], 64, 5).to(dev)
    fpn.init_weights()
    sizes = [(50, 84), (25, 42), (13, 21), (7, 11)]
    xs = [torch.randn(2, c, hh, ww, device=dev).to(torch.bfloat16).contiguous(memory_format=CL).requires_grad_()
          for c, (hh, ww) in zip([32, 64, 128, 256], sizes)]
    res = []
    for fused in (True, False):
        fpn.fused_top_down = fused
        fpn.zero_grad()
        for t in xs:
            t.grad = None
        keep = F.interpolate
        if fused:                       # the fused form must not up-sample anything
            F.interpolate = None
        try:
            outs = fpn(xs)
        finally:
            F.interpolate = keep
        assert all(o.dtype == torch.bfloat16 for o in outs)
        sum((o.float() * torch.linspace(0.5, 1.5, o.numel(), device=dev).view_as(o)).sum() for o in outs).backward()
        res.append(([o.detach().float() for o in outs], [t.grad.float() for t in xs],
                    {n: q.grad.clone() for n, q in fpn.named_parameters()}))
    (o1, g1, p1), (o2, g2, p2) = res
    for a, b in zip(o1, o2):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))
    for n in p2:
        torch.testing.assert_close(p1[n], p2[n], rtol=2e-2, atol=2e-2 * float(p2[n].abs().max()), msg=n)


@pytest.mark.gpu
@pytest.mark.parametrize('B,Ci,H,W,Co,k,stride,pad,relu,with_res', [
    (2, 64, 20, 28, 128, 3, 1, 1, True, False), (1, 256, 13, 17, 256, 1, 1, 0, False, True),
    (2, 96, 15, 15, 72, 3, 2, 1, True, True), (1, 128, 9, 40, 300, 3, 1, 2, False, False)])
def test_conv2d_bf16_forward(B, Ci, H, W, Co, k, stride, pad, relu, with_res):
    """bf16 MFMA forward (fp32 accumulate) against an fp32 convolution of the same bf16-rounded operands: the only
    differences are the summation order and the final rounding to bf16."""
    from htd_amd import dense
    torch.manual_seed(Ci + Co)
    dev = torch.device('cuda:0')
    dil = 2 if pad == 2 else 1
    x = torch.randn(B, Ci, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    b = torch.randn(Co, device=dev)
    y_ref = F.conv2d(x.float(), w.float(), b, stride, pad, dil)
    res = torch.randn_like(y_ref).to(torch.bfloat16) if with_res else None
    if with_res:
        y_ref = y_ref + res.float()
    if relu:
        y_ref = y_ref.relu()
    y = dense.conv2d_bf16(x, w, b, stride, pad, dil, relu, res)
    assert y.dtype == torch.bfloat16 and y.shape == y_ref.shape
    torch.testing.assert_close(y.float(), y_ref, rtol=1e-2, atol=2e-2)
    # exactness on integer data: every product and partial sum is representable -> bit-equal to the fp32 result
    xi = torch.randint(-2, 3, x.shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wi = torch.randint(-1, 2, w.shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    yi = dense.conv2d_bf16(xi, wi, None, stride, pad, dil)
    ri = F.conv2d(xi.float(), wi.float(), None, stride, pad, dil)
    assert torch.equal(yi.float(), ri.to(torch.bfloat16).float())


@pytest.mark.gpu
@pytest.mark.parametrize('B,Ci,H,W,Co,k,stride,pad', [(2, 64, 20, 28, 128, 3, 1, 1), (1, 256, 13, 17, 256, 1, 1, 0),
                                                      (2, 96, 15, 15, 72, 3, 2, 1), (3, 40, 9, 11, 200, 3, 1, 2)])
def test_conv2d_wgrad_bf16(B, Ci, H, W, Co, k, stride, pad):
    """bf16 weight gradient (operands through the transposing LDS read) against autograd on the same bf16-rounded
    operands in fp32, and bit-exactly on small-integer data (asymmetric: catches transposed or permuted operands)."""
    from htd_amd import dense
    torch.manual_seed(Ci * 3 + Co)
    dev = torch.device('cuda:0')
    dil = 2 if pad == 2 else 1
    x = torch.randn(B, Ci, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(Co, Ci, k, k, device=dev, requires_grad=True)
    y = F.conv2d(x.float(), w, None, stride, pad, dil)
    gy = torch.randn_like(y).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref, = torch.autograd.grad(y, w, gy.float())
    got = dense.conv2d_wgrad_bf16(x, gy, w.shape, stride, pad, dil)
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    xi = torch.randint(-2, 3, x.shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gi = torch.randint(-1, 2, gy.shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    yi = F.conv2d(xi.float(), w, None, stride, pad, dil)
    refi, = torch.autograd.grad(yi, w, gi.float())
    assert torch.equal(dense.conv2d_wgrad_bf16(xi, gi, w.shape, stride, pad, dil), refi)


@pytest.mark.gpu
@pytest.mark.parametrize('Ci,Co,k,stride,pad,relu', [(64, 128, 3, 1, 1, True), (256, 64, 1, 1, 0, False),
                                                     (96, 96, 3, 2, 1, True)])
def test_conv2d_bf16_autograd_matches_fp32_autograd(Ci, Co, k, stride, pad, relu):
    """Conv2dBf16Function (bf16 activations, fp32 master weights) against fp32 autograd on the bf16-rounded operands:
    output, data gradient, fp32 weight and bias gradients, to bf16 accuracy."""
    from htd_amd import dense
    torch.manual_seed(Ci + 7 * Co)
    dev = torch.device('cuda:0')
    x = torch.randn(2, Ci, 18, 22, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, k, k, device=dev) / (Ci * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Co, device=dev) * 0.1
    xa, wa, ba = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = dense.conv2d_bf16_autograd(xa, wa, ba, stride, pad, 1, relu)
    xr = x.float().requires_grad_()
    wr = w.to(torch.bfloat16).float().requires_grad_()
    br = b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, stride, pad)
    yr = yr.relu() if relu else yr
    g = torch.randn_like(yr).to(torch.bfloat16)
    y.backward(g)
    yr.backward(g.float())
    tol = dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(y.float(), yr, **tol)
    torch.testing.assert_close(xa.grad.float(), xr.grad, rtol=3e-2, atol=3e-2 * float(xr.grad.abs().max()))
    torch.testing.assert_close(wa.grad, wr.grad, rtol=3e-2, atol=3e-2 * float(wr.grad.abs().max()))
    torch.testing.assert_close(ba.grad, br.grad, rtol=2e-2, atol=2e-2 * float(br.grad.abs().max()))
    assert wa.grad.dtype == torch.float32 and xa.grad.dtype == torch.bfloat16


@pytest.mark.gpu
def test_res_stage_bf16_fused_matches_layerwise_bf16():
    """ResLayer on bf16 activations: the one-node stage (masks / identity sum in the dgrad epilogues, prepared weight
    operands, column-sum bias gradients) against the block-by-block bf16 path -- same forward kernels, so outputs are
    identical; gradients agree within bf16 rounding of the intermediate sums."""
    import torch.nn as nn
    from htd_amd.detector.resnet import Bottleneck, ResLayer
    torch.manual_seed(3)
    dev = torch.device('cuda:0')
    for stride in (1, 2):
        layer = ResLayer(Bottleneck, 64, 32, 3, stride=stride, norm_cfg=dict(type='BN', requires_grad=True)).to(dev)
        for m in layer.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
        layer.eval()
        x = torch.randn(2, 64, 18, 22, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res, g = [], None
        for fused in (True, False):
            layer.zero_grad()
            xi = x.clone().requires_grad_()
            y = layer(xi) if fused else nn.Sequential.forward(layer, xi)
            g = torch.randn_like(y) if g is None else g
            y.backward(g)
            res.append((y.detach().float(), xi.grad.float(), {n: p.grad.clone() for n, p in layer.named_parameters()}))
        (y1, gx1, p1), (y2, gx2, p2) = res
        assert torch.equal(y1, y2)
        assert float((gx1 - gx2).norm() / gx2.norm()) < 2e-2
        assert set(p1) == set(p2)
        for n in p2:
            assert float((p1[n] - p2[n]).norm() / (p2[n].norm() + 1e-6)) < 3e-2, n


@pytest.mark.gpu
@pytest.mark.parametrize('rows,C', [(1, 4), (37, 64), (5000, 576), (268800, 64), (4200, 2048)])
def test_colsum_bf16_and_weight_prep(rows, C):
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.randn(1, C, rows, 1, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    got = dense._colsum_bf16_raw(g)
    want = g.float().sum((0, 2, 3))
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-3 * max(1.0, rows ** 0.5))
    if rows == 37:
        w = torch.randn(96, 64, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        wb, wT = dense._prep_bf16(w)
        assert torch.equal(wb, w.to(torch.bfloat16))
        assert torch.equal(wT, w.to(torch.bfloat16).flip(2, 3).permute(1, 0, 2, 3))


@pytest.mark.gpu
def test_balanced_tail_epilogue_mask_and_accum():
    """A layer size that takes the balanced-tail path (1060 tiles of 64x64: 1024 whole + 36 cut along K): the data
    gradient with the ReLU mask and the identity-branch sum in its epilogue equals mask * (plain data gradient + accum)."""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    g = torch.randn(1, 256, 130, 130, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 64, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    shape = (1, 64, 130, 130)
    mask_src = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    accum = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    plain = dense._dgrad_raw(g, w, shape, 1, 1, 1)
    fused = dense._dgrad_raw(g, w, shape, 1, 1, 1, mask_src=mask_src, accum=accum)
    torch.testing.assert_close(fused, (plain + accum) * (mask_src > 0), rtol=1e-5, atol=1e-5)
    ref = torch.nn.functional.conv_transpose2d(g.double(), w.double(), None, 1, 1)
    torch.testing.assert_close(plain.double(), ref, rtol=1e-4, atol=1e-3)


# Round 4: the bounds of round 2 again (rms <= 1.5 x the native fp32 MFMA's error, largest error <= 2 x).  They were widened to 2 x /
# 3 x in round 3 after one run read 1.89 x on the 48 -> 96 3x3 data gradient; the kernels at the end of round 3 and since measure
# 0.48-1.22 x (rms) and 0.41-1.42 x (largest) over the four cases x three passes (printed below; profiles/r04_x3_accuracy.log
# holds the five-seed table: 3x3 layers 0.61-1.09 x rms on either MFMA shape) -- the excursion belonged to an intermediate state
# of conv_x3p_kernel's K loop, not to the arithmetic.
RMS_BOUND, MAX_BOUND = float(__import__('os').environ.get('X3_RMS_BOUND', '1.5')), float(__import__('os').environ.get('X3_MAX_BOUND', '2.0'))


def _wide(shape, g, spread):
    return torch.randn(shape, generator=g) * torch.exp(torch.randn(shape, generator=g) * spread)


@pytest.mark.parametrize('Ci,Co,k,H,W', [(256, 256, 3, 40, 56), (1024, 256, 1, 50, 84), (64, 256, 1, 60, 80), (48, 96, 3, 33, 47)])
def test_split_bf16_products_are_fp32_accurate(Ci, Co, k, H, W):
    """htd_conv2d_set_math(1): fp32 products through exact three-way bf16 splits (six bf16 MFMAs per 16 k, fp32
    accumulation) against math 0 (the fp32-input MFMA) and an fp64 reference, on data with a wide dynamic range, for
    forward, data gradient and weight gradient.  Error = |result - fp64| / accumulated magnitude (sum of |products|).
      * absolute: the largest error of BOTH arithmetics stays under 1.5e-7 sqrt(K) -- fp32 rounding of a K-term sum;
      * relative: the split form sits in the error class of the native fp32 MFMA: rms error <= 2x, largest error <= 3x
        (+ 2e-8).  Measured over five seeds (tools/x3_accuracy.py, profiles/r03_x3_accuracy.log): rms 0.84-0.85x native
        on 1x1 layers (fewer roundings: six per 16 k instead of sixteen), 1.20-1.45x on 3x3 layers, whose (channel slice,
        filter row, tap) summation order differs from the native kernel's; largest error 0.6-1.9x depending on the seed
        -- the maximum over 10^5 heavy-tailed outputs is a noisy statistic, which is why the class test is on the rms."""
    from htd_amd import capi, dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(Ci + k)
    x = _wide((2, Ci, H, W), g, 2.0).to(dev).contiguous(memory_format=torch.channels_last)
    w = (_wide((Co, Ci, k, k), g, 1.0).to(dev) / (Ci * k * k) ** 0.5).contiguous(memory_format=torch.channels_last)
    p = k // 2
    xd, wd = x.double(), w.double()
    ref = F.conv2d(xd, wd, None, 1, p)
    scale = F.conv2d(xd.abs(), wd.abs(), None, 1, p)       # accumulated magnitude
    gy = torch.randn(ref.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gref = torch.nn.grad.conv2d_input(x.shape, wd, gy.double(), 1, p)
    gscale = torch.nn.grad.conv2d_input(x.shape, wd.abs(), gy.double().abs(), 1, p)
    wref = torch.nn.grad.conv2d_weight(xd, w.shape, gy.double(), 1, p)
    wscale = torch.nn.grad.conv2d_weight(xd.abs(), w.shape, gy.double().abs(), 1, p)
    L = capi.lib()
    prev = L.htd_conv2d_set_math(-1)
    err = {}
    try:
        for mode in (0, 1):
            L.htd_conv2d_set_math(mode)
            dense.new_step()
            xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
            y = dense.conv2d(xr, wr, None, 1, p, 1)
            y.backward(gy)
            e = [(y.detach().double() - ref).abs() / scale, (xr.grad.double() - gref).abs() / gscale,
                 (wr.grad.double() - wref).abs() / wscale]
            err[mode] = [(float(t.max()), float(t.pow(2).mean().sqrt())) for t in e]
    finally:
        L.htd_conv2d_set_math(prev)
        dense.new_step()
    npix = ref.shape[0] * ref.shape[2] * ref.shape[3]
    for i, (what, K) in enumerate((('forward', Ci * k * k), ('data gradient', Co * k * k), ('weight gradient', npix))):
        (max0, rms0), (max1, rms1) = err[0][i], err[1][i]
        print(f'{what:15s} K={K:6d}: split-bf16 / native  rms {rms1 / rms0:.2f}  max {max1 / max0:.2f}')
        assert max0 < 1.5e-7 * K ** 0.5 and max1 < 1.5e-7 * K ** 0.5, (what, err, K)
        assert rms1 <= RMS_BOUND * rms0, (what, err, K)
        assert max1 <= MAX_BOUND * max0 + 2e-8, (what, err, K)


X3P_CASES = [
    # B, Ci, H, W, Co, k, stride: layers conv_x3p_kernel takes (csrc/conv_x3.hip) -- borders, ragged tiles, narrow maps whose
    # halo runs wrap over several rows and images, Co beyond the last tile, strided 1x1, split-K lengths
    (2, 64, 20, 28, 128, 3, 1), (1, 256, 13, 17, 256, 3, 1), (5, 576, 7, 7, 576, 3, 1), (37, 32, 7, 7, 96, 3, 1),
    (1, 16, 1, 1, 64, 3, 1), (1, 16, 3, 200, 40, 3, 1), (2, 48, 5, 3, 33, 3, 1), (1, 32, 130, 130, 256, 3, 1),
    (2, 64, 20, 28, 64, 1, 1), (2, 256, 20, 28, 512, 1, 2), (3, 32, 9, 11, 576, 1, 3), (1, 12544, 37, 1, 1024, 1, 1),
    (1, 1024, 200, 1, 81, 1, 1),
]


@pytest.mark.parametrize('B,Ci,H,W,Co,k,stride', X3P_CASES)
def test_x3p_kernel_is_exact_on_integers_and_matches_the_igemm_kernel(B, Ci, H, W, Co, k, stride):
    """conv_x3p_kernel (weights pre-split into bf16 planes once per step, activations staged as halo runs): on small-integer
    data every product and partial sum is exact in fp32, so forward and data gradient must equal the fp64 convolution BIT FOR
    BIT whatever the summation order -- any wrong tap, border mask, halo row or weight-plane chunk shows.  On real data it
    agrees with conv_igemm_kernel<X3> (same products, other summation order) to fp32 rounding, with every epilogue operand
    (bias, residual, ReLU; mask_src and accum in the data gradient)."""
    import os
    from htd_amd import capi, dense
    L = capi.lib()
    p = k // 2
    if not L.htd_conv2d_x3p_supported(Ci, Co, k, k, stride, p, 1):
        pytest.skip('conv_x3p_kernel switched off (HTD_X3P=0 / HTD_CONV_MATH=0)')
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B + Ci + Co + H)
    xi = torch.randint(-4, 5, (B, Ci, H, W), generator=g).float()
    wi = torch.randint(-3, 4, (Co, Ci, k, k), generator=g).float()
    ref = F.conv2d(xi.double(), wi.double(), None, stride, p)
    xd = xi.to(dev).contiguous(memory_format=CL)
    wd = wi.to(dev).contiguous(memory_format=CL)
    dense.new_step()
    y = dense._fwd_raw(xd, wd, None, None, stride, p, 1, False)
    assert torch.equal(y.cpu().double(), ref)
    if stride == 1 and L.htd_conv2d_x3p_supported(Co, Ci, k, k, 1, p, 1):
        gi = torch.randint(-4, 5, tuple(ref.shape), generator=g).float()
        gref = torch.nn.grad.conv2d_input(xi.shape, wi.double(), gi.double(), 1, p)
        gx = dense._dgrad_raw(gi.to(dev).contiguous(memory_format=CL), wd, xd.shape, 1, p, 1)
        assert torch.equal(gx.cpu().double(), gref)
    # real data, all epilogue operands, against the other kernel (direct C-ABI call of htd_conv2d_fwd / _bwd_data)
    x = torch.randn(B, Ci, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).to(dev).contiguous(memory_format=CL)
    b = torch.randn(Co, generator=g).to(dev)
    r = torch.randn(tuple(ref.shape), generator=g).to(dev).contiguous(memory_format=CL)
    y = dense._fwd_raw(x, w, b, r, stride, p, 1, True)
    Ho, Wo = ref.shape[2], ref.shape[3]
    y0 = torch.empty_like(y)
    P, S = capi.ptr, capi.current_stream_ptr
    capi.call('htd_conv2d_fwd', P(x), P(w), P(b), P(r), 0, 0, P(y0), B, H, W, Ci, Co, k, k, stride, p, 1, 1,
              P(dense._splitk_ws(B * Ho * Wo, Co, Ci, k, k, dev)), S())
    torch.testing.assert_close(y, y0, rtol=2e-5, atol=2e-6 * (Ci * k * k) ** 0.5)
    if stride == 1 and L.htd_conv2d_x3p_supported(Co, Ci, k, k, 1, p, 1):
        gy = torch.randn_like(y)
        acc = torch.randn_like(x)
        gx = dense._dgrad_raw(gy, w, x.shape, 1, p, 1, mask_src=x, accum=acc)
        wT = torch.empty(w.numel(), device=dev)
        capi.call('htd_conv2d_flip_weights', P(w), P(wT), Co, k, k, Ci, S())
        gx0 = torch.empty_like(gx)
        capi.call('htd_conv2d_bwd_data', P(gy), P(wT), P(x), P(acc), P(gx0), B, H, W, Ci, Co, k, k, 1, p, 1,
                  P(dense._splitk_ws(B * H * W, Ci, Co, k, k, dev)), S())
        torch.testing.assert_close(gx, gx0, rtol=2e-5, atol=2e-6 * (Co * k * k) ** 0.5)
    dense.new_step()


def test_x3p_weight_planes_follow_the_optimizer_kernel():
    """the per-step caches of derived weight images are keyed on tensor._version, which the ctypes optimizer
    kernel (mmcv_ops.sgd_momentum_step_) does not bump.  The plane cache also carries mmcv_ops.PARAM_EPOCH: a convolution
    after such an update, with no dense.new_step() in between, must see the new weights (forward and data gradient)."""
    from htd_amd import dense
    from htd_amd import mmcv_ops as M
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    x = torch.randn(2, 64, 12, 14, device=dev).contiguous(memory_format=CL)
    gy = torch.randn(2, 64, 12, 14, device=dev).contiguous(memory_format=CL)
    wflat = torch.randn(64 * 3 * 3 * 64, device=dev) * 0.05            # KRSC memory, updated in place by the kernel
    w = wflat.view(64, 3, 3, 64).permute(0, 3, 1, 2)
    assert w.is_contiguous(memory_format=CL)
    dense.new_step()
    dense._fwd_raw(x, w, None, None, 1, 1, 1, False)
    dense._dgrad_raw(gy, w, x.shape, 1, 1, 1)
    grad, mom = torch.randn_like(wflat), torch.zeros_like(wflat)
    M.sgd_momentum_step_(wflat, grad, mom, torch.tensor([0.5], device=dev), 0.9, 0.0)
    y = dense._fwd_raw(x, w, None, None, 1, 1, 1, False)
    gx = dense._dgrad_raw(gy, w, x.shape, 1, 1, 1)
    torch.testing.assert_close(y.cpu().double(), F.conv2d(x.cpu().double(), w.cpu().double(), None, 1, 1), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gx.cpu().double(), torch.nn.grad.conv2d_input(x.shape, w.cpu().double(), gy.cpu().double(), 1, 1),
                               rtol=1e-4, atol=1e-4)
    dense.new_step()


@pytest.mark.gpu
def test_chained_conv_consumers_hand_one_gradient_to_the_producer():
    """Conv2dFunction(chain=True): the second reader of a map reads the alias the first one returns; in backward its
    gradient joins in the first reader's data-gradient epilogue.  Same gradients as two independent readers."""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 32, 20, 28, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wa = (torch.randn(48, 32, 3, 3, generator=g) * 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    wb = (torch.randn(16, 32, 1, 1, generator=g) * 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    ra, rb = torch.randn(2, 48, 20, 28, generator=g).to(dev), torch.randn(2, 16, 20, 28, generator=g).to(dev)

    def run(chain):
        x = x0.clone().requires_grad_()
        a, b = wa.clone().requires_grad_(), wb.clone().requires_grad_()
        h = x * 1.0                                   # a produced map (not a leaf): its gradient is what gets summed
        if chain:
            ya, alias = dense.conv2d(h, a, None, 1, 1, 1, chain=True)
            yb = dense.conv2d(alias, b, None, 1, 0, 1)
        else:
            ya, yb = dense.conv2d(h, a, None, 1, 1, 1), dense.conv2d(h, b, None, 1, 0, 1)
        ((ya * ra).sum() + (yb * rb).sum()).backward()
        return x.grad, a.grad, b.grad
    ref, out = run(False), run(True)
    for r, o in zip(ref, out):
        assert torch.allclose(r, o, rtol=1e-5, atol=1e-5 * float(r.abs().max()))
    # the alias alone (first reader's output unused) still carries the second reader's gradient
    x = x0.clone().requires_grad_()
    ya, alias = dense.conv2d(x * 1.0, wa, None, 1, 1, 1, chain=True)
    (dense.conv2d(alias, wb, None, 1, 0, 1) * rb).sum().backward()
    assert torch.allclose(x.grad, ref[0] - run_only_a(x0, wa, ra), rtol=1e-4, atol=1e-5 * float(ref[0].abs().max()))


def run_only_a(x0, wa, ra):
    from htd_amd import dense
    x = x0.clone().requires_grad_()
    (dense.conv2d(x * 1.0, wa, None, 1, 1, 1) * ra).sum().backward()
    return x.grad


@pytest.mark.gpu
def test_step_flip_cache_never_serves_a_stale_or_misshapen_image():
    """dense.flip_many / the per-step flip cache of _dgrad_raw: one storage read as two weight shapes gets two images, an
    in-place weight update (version bump) or new_step() invalidates, and the cached result equals a fresh flip."""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    CL = torch.channels_last
    g = torch.Generator().manual_seed(9)
    lin = (torch.randn(64, 32 * 9, generator=g) * 0.1).to(dev)                   # a Linear weight ...
    as_conv = lin.view(64, 3, 3, 32).permute(0, 3, 1, 2)                          # ... and the same storage as a 3x3 conv (KRSC)
    assert as_conv.data_ptr() == lin.data_ptr() and as_conv.is_contiguous(memory_format=CL)
    gy_lin = torch.randn(5, 64, 1, 1, generator=g).to(dev).contiguous(memory_format=CL)
    gy_conv = torch.randn(2, 64, 6, 6, generator=g).to(dev).contiguous(memory_format=CL)

    def fresh(gy, w, xs, pad):
        dense.new_step()
        return dense._dgrad_raw(gy, w, xs, 1, pad, 1)
    ref_lin = fresh(gy_lin, lin.view(64, 288, 1, 1), (5, 288, 1, 1), 0)
    ref_conv = fresh(gy_conv, as_conv, (2, 32, 6, 6), 1)
    dense.new_step()
    dense.flip_many([lin])                                                        # registers the (64, 288, 1, 1) reading only
    assert torch.equal(dense._dgrad_raw(gy_lin, lin.view(64, 288, 1, 1), (5, 288, 1, 1), 1, 0, 1), ref_lin)
    assert torch.equal(dense._dgrad_raw(gy_conv, as_conv, (2, 32, 6, 6), 1, 1, 1), ref_conv)       # other shape: its own flip
    assert torch.equal(dense._dgrad_raw(gy_conv, as_conv, (2, 32, 6, 6), 1, 1, 1), ref_conv)       # now from the cache
    lin.mul_(2.0)                                                                 # in-place update: version changes
    assert torch.allclose(dense._dgrad_raw(gy_conv, as_conv, (2, 32, 6, 6), 1, 1, 1), 2.0 * ref_conv, rtol=1e-6, atol=1e-6)
    dense.new_step()
    assert len(dense._STEP_FLIPS) == 0


WGRAD_CASES = [
    # B, Ci, H, W, Co, k, stride, pad: what conv_wgrad_x3d_kernel (1x1, strided, 7x7) and conv_wgrad_x3hd_kernel (3x3 stride 1,
    # W + 1 >= 16) take -- ragged channel counts, K ranges that end inside a slice, maps one padding column wide of a slice,
    # splits with an odd number of 16-pixel slices, a single image row
    (2, 64, 20, 28, 128, 1, 1, 0), (3, 100, 9, 11, 132, 1, 1, 0), (1999, 36, 1, 1, 68, 1, 1, 0), (2, 256, 20, 28, 512, 1, 2, 0),
    (2, 8, 64, 96, 64, 7, 2, 3), (2, 128, 40, 56, 128, 3, 2, 1), (2, 64, 20, 28, 128, 3, 1, 1), (2, 36, 9, 17, 68, 3, 1, 1),
    (3, 132, 5, 15, 200, 3, 1, 1), (1, 64, 1, 40, 64, 3, 1, 1), (4, 256, 13, 21, 256, 3, 1, 1), (1, 32, 130, 130, 256, 3, 1, 1),
    # narrow maps, where a 16-pixel slice spans several image rows (the RoI regression branch: 7x7), down to maps too small for
    # the interleaved kernel (2x3: the phased one takes them)
    (24, 576, 7, 7, 576, 3, 1, 1), (37, 32, 7, 7, 96, 3, 1, 1), (2, 48, 5, 3, 64, 3, 1, 1), (5, 64, 2, 3, 64, 3, 1, 1),
    (3, 64, 9, 13, 128, 3, 1, 1),
    # 64-wide tiles of conv_wgrad_x3d_kernel: Co <= 64 (64 x 128), Ci <= 64 on a 1x1 layer (128 x 64), ragged forms of both
    (2, 256, 20, 28, 64, 1, 1, 0), (2, 64, 20, 28, 256, 1, 1, 0), (3, 36, 9, 11, 40, 1, 1, 0), (3, 40, 9, 11, 132, 1, 1, 0),
    (2, 16, 33, 47, 48, 3, 2, 1),
]
_WGRAD_CHILD = r'''
import sys, zlib, torch
sys.path.insert(0, sys.argv[1])
from htd_amd import capi, dense
cases = eval(sys.argv[2])
dev = torch.device('cuda', 0)
for B, Ci, H, W, Co, k, s, p in cases:
    g = torch.Generator().manual_seed(B + Ci + Co)
    x = torch.randn(B, Ci, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    gy = torch.randn(B, Co, Ho, Wo, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.empty(Co, Ci, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    gw, gb = dense._wgrad_launch(x, gy, w, s, p, 1, True)[:2]
    print(zlib.crc32(gw.cpu().contiguous(memory_format=torch.channels_last).numpy().tobytes()), zlib.crc32(gb.cpu().numpy().tobytes()),
          ','.join(v.hex() for v in gb.cpu().double().tolist()))
'''


def test_interleaved_weight_gradient_kernels_match_the_phased_ones_bit_for_bit():
    """conv_wgrad_x3d_kernel / conv_wgrad_x3hd_kernel (csrc/conv_wgrad.hip: split of the next slice between the MFMAs of the
    current one, buffer-descriptor loads) sum in the order of conv_wgrad_x3_kernel / conv_wgrad_x3h_kernel, which
    HTD_WGRAD_X3D=0 selects: weight and bias gradients of both must be the same bits.  One exception: layers with Co <= 64 run
    on 64-row tiles (both kernels), whose 16 staging rows per slice (instead of 2 x 8) associate the bias-gradient partial sums differently --
    there the weight gradient is still the same bits and the bias gradient agrees to fp32 rounding.  The switch is read once
    per process, so each side runs in a child process (two GPU processes, one after the other)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for flag in ('1', '0'):
        env = dict(os.environ, HTD_WGRAD_X3D=flag)
        r = subprocess.run([sys.executable, '-c', _WGRAD_CHILD, root, repr(WGRAD_CASES)], env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[flag] = [l for l in r.stdout.splitlines() if l and l[0].isdigit()]
        assert len(out[flag]) == len(WGRAD_CASES), r.stdout
    for case, a, b in zip(WGRAD_CASES, out['1'], out['0']):
        (wa, ba, va), (wb, bb, vb) = a.split(), b.split()
        assert wa == wb, (case, wa, wb)
        if case[4] > 64:                                     # Co > 64: 128-row tiles in both kernels
            assert ba == bb, (case, ba, bb)
        else:
            ga = torch.tensor([float.fromhex(v) for v in va.split(',')], dtype=torch.float64)
            gb = torch.tensor([float.fromhex(v) for v in vb.split(',')], dtype=torch.float64)
            assert float((ga - gb).abs().max()) <= 2e-6 * float(gb.abs().max()), (case, float((ga - gb).abs().max()))


@pytest.mark.parametrize('B,Ci,H,W,Co,k,stride,pad', WGRAD_CASES)
def test_weight_gradient_is_exact_on_integers(B, Ci, H, W, Co, k, stride, pad):
    """Small-integer operands: every product and partial sum is exactly representable, so the split-bf16 weight gradient
    (and the bias gradient of the same launch) must EQUAL the fp64 reference -- any dropped / doubled row, column or tap of
    the staging (slice ends, padding column, ragged tiles, buffer-descriptor bounds) shows as a wrong integer."""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(7 * B + Ci)
    x = torch.randint(-4, 5, (B, Ci, H, W), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    gy = torch.randint(-4, 5, (B, Co, Ho, Wo), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.empty(Co, Ci, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    gw, gb = dense._wgrad_launch(x, gy, w, stride, pad, 1, True)[:2]
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), stride, pad)
    assert torch.equal(gw.double(), ref)
    assert torch.equal(gb.double(), gy.double().sum((0, 2, 3)))


@pytest.mark.parametrize('B,H,W,relu', [(2, 64, 96, True), (1, 33, 47, False), (3, 7, 301, True), (1, 1, 1, False), (2, 130, 258, True)])
def test_stem7_kernel_matches_fp64_and_is_exact_on_integers(B, H, W, relu):
    """htd_conv2d_stem7_fwd (csrc/conv_stem.hip: the 7x7 / stride 2 / padding 3 stem on a 4-channel image, reduction over
    filter rows) against an fp64 convolution: odd sizes (output rows that end inside a 128-pixel tile, maps narrower than the
    filter), borders on every side; integer operands must give the exact integers; random operands stay within fp32
    rounding of the accumulated magnitude (the bound of test_split_bf16_products_are_fp32_accurate).  Through dense.conv2d the
    3-channel image takes this path and the weight gradient the 4-channel form of conv_wgrad_x3d_kernel."""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    xi = torch.randint(-3, 4, (B, 3, H, W), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last)
    wi = torch.randint(-3, 4, (64, 3, 7, 7), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last)
    bi = torch.randint(-3, 4, (64, ), generator=g).float().to(dev)
    y = dense.conv2d(xi, wi, bi, 2, 3, 1, relu=relu)
    ref = F.conv2d(xi.double(), wi.double(), bi.double(), 2, 3)
    ref = ref.clamp(min=0) if relu else ref
    assert torch.equal(y.double(), ref)
    x = torch.randn(B, 3, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = torch.randn(64, generator=g).to(dev).requires_grad_()
    y = dense.conv2d(x, w, b, 2, 3, 1, relu=relu)
    ref = F.conv2d(x.double(), w.detach().double(), b.detach().double(), 2, 3)
    scale = F.conv2d(x.double().abs(), w.detach().double().abs(), b.detach().double().abs(), 2, 3)
    ref = ref.clamp(min=0) if relu else ref
    assert float(((y.detach().double() - ref).abs() / scale).max()) < 1.5e-7 * 148 ** 0.5
    gy = torch.randn(y.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    gm = gy.double() * (ref > 0) if relu else gy.double()
    wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gm, 2, 3)
    wscale = torch.nn.grad.conv2d_weight(x.double().abs(), w.shape, gm.abs(), 2, 3) + 1e-30
    npix = y.shape[0] * y.shape[2] * y.shape[3]
    assert float(((w.grad.double() - wref).abs() / wscale).max()) < 1.5e-7 * max(npix, 1) ** 0.5 + 1e-12
    assert float((b.grad.double() - gm.sum((0, 2, 3))).abs().max()) <= 1e-5 * float(gm.abs().sum((0, 2, 3)).max()) + 1e-12


@pytest.mark.parametrize('tool,n', [('fuzz_conv.py', 40), ('fuzz_wgrad.py', 40)])
def test_random_convolution_shapes_are_exact_on_integers(tool, n):
    """tools/fuzz_conv.py: random problems through dense.conv2d and its backward (forward / data-gradient kernels with their
    epilogues, ReLU masks, weight gradients) on small-integer operands must equal the fp64 reference; tools/fuzz_wgrad.py: random
    weight-gradient problems, interleaved kernels = phased kernels bit for bit (its own child process), and random image sizes
    through the stem kernel.  A fixed seed here; profiles/r03_fuzz_*.log hold the longer runs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', tool), str(n), '11'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert '0 mismatches' in r.stdout, r.stdout[-2000:]
