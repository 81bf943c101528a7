"""Deformable conv (v1 'DCN' and v2) kernels against the CPU oracle (C forward, torch autograd restatement)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


@pytest.mark.parametrize('B,C,H,W,Co,stride,with_mask', [(2, 32, 11, 13, 24, 1, False), (1, 64, 9, 10, 32, 2, False),
                                                          (2, 16, 8, 8, 16, 1, True),
                                                          (2, 128, 19, 21, 24, 1, False),     # tiled col2im, 2 slices
                                                          (1, 64, 17, 9, 16, 1, True), (2, 192, 13, 12, 8, 2, False)])
def test_deform_conv_fwd_bwd(B, C, H, W, Co, stride, with_mask):
    from htd_amd.dcn import deform_conv2d
    from oracle import ops as O
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    off = torch.randn(B, 18, Ho, Wo, generator=g) * 1.5           # up to several pixels, incl. out-of-image taps
    mask = torch.rand(B, 9, Ho, Wo, generator=g) if with_mask else None
    ref_c = O.deform_conv2d(x, off, w, stride, 1, 1, mask=mask)
    xr, offr, wr = x.clone().requires_grad_(), off.clone().requires_grad_(), w.clone().requires_grad_()
    mr = mask.clone().requires_grad_() if with_mask else None
    ref = O.deform_conv2d_autograd(xr, offr, wr, stride, 1, 1, mask=mr)
    torch.testing.assert_close(ref.detach(), ref_c, rtol=1e-4, atol=1e-4)      # the two oracle forms agree
    xd = x.to(dev).contiguous(memory_format=CL).requires_grad_()
    od = off.to(dev).contiguous(memory_format=CL).requires_grad_()
    wd = w.to(dev).contiguous(memory_format=CL).requires_grad_()
    md = mask.to(dev).contiguous(memory_format=CL).requires_grad_() if with_mask else None
    y = deform_conv2d(xd, od, wd, stride, 1, 1, mask=md)
    torch.testing.assert_close(y.cpu(), ref_c, rtol=1e-4, atol=1e-4)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    y.backward(go.to(dev))
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(od.grad.cpu(), offr.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-3, atol=1e-3)
    if with_mask:
        torch.testing.assert_close(md.grad.cpu(), mr.grad, rtol=1e-3, atol=1e-3)


def test_zero_offsets_equal_plain_conv_and_pack_layer():
    """conv_offset is zero-initialised (resnet.py:608-612): a fresh DCN layer is exactly the plain 3x3 conv."""
    from htd_amd.registry import CONV_LAYERS
    from htd_amd import detector  # noqa: F401
    dev = torch.device('cuda:0')
    layer = CONV_LAYERS.get('DCN')(32, 48, kernel_size=3, stride=1, padding=1, dilation=1, bias=False, deform_groups=1).to(dev)
    assert hasattr(layer, 'conv_offset') and layer.conv_offset.weight.abs().sum().item() == 0.0
    x = torch.randn(2, 32, 10, 12, device=dev).contiguous(memory_format=CL)
    y = layer(x)
    ref = F.conv2d(x.cpu().double(), layer.weight.detach().cpu().double(), None, 1, 1)
    torch.testing.assert_close(y.detach().cpu().double(), ref, rtol=1e-4, atol=1e-4)
    v2 = CONV_LAYERS.get('DCNv2')(32, 48, kernel_size=3, stride=1, padding=1, dilation=1, deform_groups=1).to(dev)
    y2 = v2(x)                                              # mask = sigmoid(0) = 0.5
    ref2 = 0.5 * F.conv2d(x.cpu().double(), v2.weight.detach().cpu().double(), None, 1, 1)
    torch.testing.assert_close(y2.detach().cpu().double(), ref2, rtol=1e-4, atol=1e-4)


def test_r101_dcn_backbone_block_matches_oracle():
    """One DCN bottleneck (conv2 = DeformConv2dPack with non-zero offsets) against oracle.bottleneck."""
    from golden_util import seeded_state_value
    from htd_amd.detector.resnet import Bottleneck, ResLayer
    from oracle import detector as D
    dev = torch.device('cuda:0')
    layer = ResLayer(Bottleneck, 64, 32, 1, stride=2, dcn=dict(type='DCN', deform_groups=1, fallback_on_stride=False),
                     norm_cfg=dict(type='BN', requires_grad=True)).eval()
    sd = {}
    with torch.no_grad():
        for k, t in layer.state_dict().items():
            if k.endswith('num_batches_tracked'):
                continue
            v = torch.from_numpy(seeded_state_value('dcnblk.' + k, t.shape))
            if 'conv_offset' in k:
                v = v * 3.0
            t.copy_(v)
            sd['blk.' + k[2:]] = v.clone()
    x = torch.randn(2, 64, 12, 14, generator=torch.Generator().manual_seed(1))
    ref = D.bottleneck(sd, 'blk', x, 2, dcn=True)
    y = layer.to(dev)(x.to(dev).contiguous(memory_format=CL))
    torch.testing.assert_close(y.detach().cpu(), ref, rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
def test_dcn_bottleneck_folded_bn_relu_epilogue():
    """Bottleneck with a deformable conv2: frozen BN + ReLU folded into the epilogue of the DCN GEMM against the
    unfused relu(bn(dcn(x))) of the reference (backbones/resnet.py:278-291), forward and every gradient."""
    import torch.nn as nn
    from htd_amd.detector.resnet import Bottleneck
    torch.manual_seed(2)
    dev = torch.device('cuda:0')
    blk = Bottleneck(64, 16, dcn=dict(type='DCN', deform_groups=1, fallback_on_stride=False)).to(dev)
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    blk.conv2.conv_offset.weight.data.normal_(0, 0.05)        # non-trivial offsets
    blk.conv2.conv_offset.bias.data.normal_(0, 0.5)
    blk.eval()
    x = torch.randn(2, 64, 14, 18, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    res, g = [], None
    for fused in (True, False):
        blk.fuse_dcn_bn = fused
        blk.zero_grad()
        x.grad = None
        y = blk(x)
        g = torch.randn_like(y) if g is None else g
        y.backward(g)
        res.append((y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()}))
    (y1, gx1, p1), (y2, gx2, p2) = res
    torch.testing.assert_close(y1, y2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx1, gx2, rtol=1e-4, atol=1e-5)
    assert set(p1) == set(p2)
    for n in p2:
        torch.testing.assert_close(p1[n], p2[n], rtol=1e-4, atol=1e-5 * max(1.0, float(p2[n].abs().max())), msg=n)


def test_col2im_row_owned_kernel_matches_direct_kernel():
    """htd_deform_col2im picks the row-owned LDS kernel for wide layers; the direct scatter kernel (taken when gx is
    not requested... or for narrow layers) is the reference here: same gx within float summation order, identical
    offset gradients, for small and for large offsets (beyond the LDS window margin), stride 1 and 2, and with
    non-finite gradients staying inside the buffers."""
    from htd_amd import capi
    P, S = capi.ptr, capi.current_stream_ptr
    dev = torch.device('cuda:0')
    for (B, H, W, C, stride, std) in [(2, 37, 45, 128, 1, 0.4), (1, 40, 33, 64, 2, 0.4), (2, 21, 19, 192, 1, 4.0)]:
        g = torch.Generator(device='cpu').manual_seed(H)
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        x = torch.randn(B, H, W, C, generator=g).to(dev)
        off = (torch.randn(B, Ho, Wo, 18, generator=g) * std).to(dev)
        gcol = torch.randn(B * Ho * Wo, 9, C, generator=g).to(dev)
        # reference: scatter each 32-channel half separately (C/2 is not a multiple of 64 for C = 64, 192; for C = 128
        # the halves are 64 wide, so use quarter slices there) -> the direct kernel
        gx = torch.zeros_like(x)
        goff = torch.empty_like(off)
        capi.call('htd_deform_col2im', P(x), P(off), None, P(gcol), P(gx), P(goff), None, B, H, W, C, 3, 3, stride, 1, 1, 1, S())
        ref_gx = torch.zeros_like(x)
        ref_goff = torch.zeros_like(off)
        w = 32
        for c0 in range(0, C, w):
            xs = x[..., c0:c0 + w].contiguous()
            gs = gcol[..., c0:c0 + w].contiguous()
            gxs = torch.zeros_like(xs)
            gos = torch.empty_like(off)
            capi.call('htd_deform_col2im', P(xs), P(off), None, P(gs), P(gxs), P(gos), None, B, H, W, w, 3, 3, stride, 1, 1,
                      1, S())
            ref_gx[..., c0:c0 + w] = gxs
            ref_goff += gos
        torch.testing.assert_close(gx, ref_gx, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(goff, ref_goff, rtol=1e-4, atol=1e-3)
    # a NaN gradient stays a NaN in gx and nothing faults
    gcol[(Ho // 2) * Wo + Wo // 2 - 3:(Ho // 2) * Wo + Wo // 2 + 3, :, 0] = float('nan')     # centre pixels, all taps
    gx.zero_()
    capi.call('htd_deform_col2im', P(x), P(off), None, P(gcol), P(gx), P(goff), None, B, H, W, C, 3, 3, stride, 1, 1, 1, S())
    torch.cuda.synchronize()
    assert torch.isnan(gx).any()


@pytest.mark.parametrize('B,C,H,W,Co,stride', [(2, 64, 13, 15, 64, 1), (1, 128, 10, 12, 96, 2)])
def test_deform_conv_bf16_tracks_fp32(B, C, H, W, Co, stride):
    """bf16 mode of the deformable conv (bf16 columns / GEMMs, fp32 offsets and accumulation) against the fp32 kernels
    on the same bf16-rounded operands: relative L2 error of the output <= 0.5 %, of the gradients <= 1-2 %."""
    from htd_amd.dcn import deform_conv2d
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(C + H)
    BF = torch.bfloat16
    x = torch.randn(B, C, H, W, generator=g).to(BF).float()
    w = (torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(BF).float()
    bias = torch.randn(Co, generator=g) * 0.1
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    off = torch.randn(B, 18, Ho, Wo, generator=g) * 1.2
    go = torch.randn(B, Co, Ho, Wo, generator=g).to(BF).float()
    names = ['y', 'gx', 'goffset', 'gw', 'gbias']
    for relu in (False, True):
        outs = []
        for dt in (torch.float32, BF):
            xd = x.to(dev).to(dt).contiguous(memory_format=CL).requires_grad_()
            od = off.to(dev).contiguous(memory_format=CL).requires_grad_()
            wd = w.to(dev).contiguous(memory_format=CL).requires_grad_()
            bd = bias.to(dev).requires_grad_()
            y = deform_conv2d(xd, od, wd, stride, 1, 1, bias=bd, relu=relu)
            assert y.dtype == dt
            y.backward(go.to(dev).to(dt))
            outs.append([t.detach().float() for t in (y, xd.grad, od.grad, wd.grad, bd.grad)])
        for n, a, b in zip(names, *outs):
            # relative L2 error.  With the ReLU a mask bit can flip where y is within rounding of 0, which moves single
            # gradient elements (and the short sums behind gbias) by a whole contribution: without it every gradient
            # is held tightly, with it the output is, and the gradients by direction
            if relu and n != 'y':
                cos = float((a * b).sum() / (a.norm() * b.norm()))
                assert cos >= 0.99, (n, cos)
                continue
            err = float((a - b).norm() / a.norm())
            tol = {'y': 5e-3, 'gx': 1.5e-2, 'goffset': 2e-2, 'gw': 1e-2, 'gbias': 5e-3}[n]
            assert err <= tol, (n, relu, err)
