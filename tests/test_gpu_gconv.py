"""Grouped 3x3 convolution kernels (ResNeXt conv2; csrc/gconv.hip) against the fp32 ATen CPU convolution -- the operator
the reference runs (backbones/resnext.py:54-63 -> nn.Conv2d(groups=64)).  fp32 tolerance 1e-4 relative."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


@pytest.mark.parametrize('C,groups,H,W,stride,dil', [
    (64, 16, 13, 17, 1, 1),      # 4 channels per group (X101 stage 1): four groups share a slab
    (64, 8, 12, 9, 2, 1),        # 8 per group, the stride-2 block of a stage
    (64, 4, 10, 11, 1, 2),       # 16 per group = one slab, dilated
    (128, 4, 9, 14, 2, 1),       # 32 per group: two input slabs per output slab
    (192, 3, 7, 8, 1, 1),        # 64 per group
    (256, 64, 20, 24, 1, 1)])    # the real stage-1 geometry
def test_grouped_conv_fwd_bwd(C, groups, H, W, stride, dil):
    from htd_amd.dense import grouped_conv2d
    g = torch.Generator().manual_seed(C + groups)
    cg = C // groups
    x = torch.randn(2, C, H, W, generator=g)
    w = torch.randn(C, cg, 3, 3, generator=g) / (cg * 9) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = torch.relu(F.conv2d(xr, wr, br, stride, dil, dil, groups))
    dev = torch.device('cuda:0')
    xd = x.to(dev).contiguous(memory_format=CL).requires_grad_()
    wd = w.to(dev).contiguous(memory_format=CL).requires_grad_()
    bd = b.to(dev).requires_grad_()
    y = grouped_conv2d(xd, wd, bd, stride, dil, dil, groups, relu=True)
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    y.backward(go.to(dev).contiguous(memory_format=CL))
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bd.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4)


def test_grouped_conv_rejects_unsupported_shapes():
    from htd_amd.dense import grouped_conv2d
    dev = torch.device('cuda:0')
    x = torch.zeros(1, 24, 8, 8, device=dev).contiguous(memory_format=CL)
    with pytest.raises(ValueError):
        grouped_conv2d(x, torch.zeros(24, 3, 3, 3, device=dev), None, 1, 1, 1, 8)        # C % 16, 3 per group
    with pytest.raises(ValueError):
        grouped_conv2d(x, torch.zeros(32, 4, 3, 3, device=dev), None, 1, 1, 1, 8)        # channel mismatch


@pytest.mark.parametrize('dcn', [False, True])
def test_resnext_stage_matches_oracle(dcn):
    """A ResNeXt stage (backbones/resnext.py: groups=8, base_width=4 -> 4 channels per group at planes=64... scaled down
    from 64x4d) with plain and deformable grouped conv2, forward and every gradient, against oracle.bottleneck."""
    from golden_util import seeded_state_value
    from htd_amd.detector.resnet import Bottleneck, ResLayer
    from oracle import detector as D
    dev = torch.device('cuda:0')
    kw = dict(dcn=dict(type='DCN', deform_groups=1, fallback_on_stride=False)) if dcn else {}
    layer = ResLayer(Bottleneck, 64, 64, 2, stride=2, groups=8, base_width=4, base_channels=64,
                     norm_cfg=dict(type='BN', requires_grad=True), **kw).eval()
    assert layer[0].width == 32 and layer[0].conv2.weight.shape == (32, 4, 3, 3)
    sd = {}
    with torch.no_grad():
        for k, t in layer.state_dict().items():
            if k.endswith('num_batches_tracked'):
                continue
            v = torch.from_numpy(seeded_state_value('xblk.' + k, t.shape))
            if 'conv_offset' in k:
                v = v * 3.0
            t.copy_(v)
            sd['blk' + k] = v.clone().requires_grad_(v.is_floating_point() and 'running' not in k)
    x = torch.randn(2, 64, 12, 14, generator=torch.Generator().manual_seed(1))
    xr = x.clone().requires_grad_()
    ref = D.bottleneck(sd, 'blk1', D.bottleneck(sd, 'blk0', xr, 2, dcn=dcn, groups=8), 1, dcn=dcn, groups=8)
    layer = layer.to(dev)
    xd = x.to(dev).contiguous(memory_format=CL).requires_grad_()
    y = layer(xd)
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-3, atol=1e-4)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(2))
    ref.backward(go)
    y.backward(go.to(dev).contiguous(memory_format=CL))
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-4)
    for k, p in layer.named_parameters():
        want = sd['blk' + k].grad
        scale = max(1.0, float(want.abs().max()))
        torch.testing.assert_close(p.grad.cpu().contiguous(), want, rtol=1e-3, atol=2e-4 * scale, msg=k)
    if not dcn:      # the plain stage also against the reference's own module (tests/golden/resnext_stage.npz)
        import numpy as np
        from golden_util import digest
        g = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'resnext_stage.npz'))
        np.testing.assert_allclose(y.detach().cpu().numpy(), g['y'], rtol=1e-3, atol=1e-4)
        for key, t in [('gx', xd.grad)] + [('grad.' + k, p.grad) for k, p in layer.named_parameters()]:
            sums, sample = digest(t.detach().cpu().contiguous())
            refs = g[key + '.sample']
            np.testing.assert_allclose(sample, refs, rtol=2e-3, atol=5e-4 * max(1.0, np.abs(refs).max()), err_msg=key)


def test_resnext101_dcn_detector_train_step():
    """htd_resnetx101_dcn_2x_mstrain.py:138-150 (ResNeXt-101 64x4d, DCN in c3-c5): state-dict keys / shapes of the
    reference module tree and two finite train steps."""
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.runner import Trainer, synthetic_batch
    from test_gpu_edge import _small
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = build_htd_detector(cfg=_small(htd_config(101, dcn=True, resnext=True)))
    sd = model.state_dict()
    assert sd['backbone.layer1.0.conv1.weight'].shape == (256, 64, 1, 1)
    assert sd['backbone.layer1.0.conv2.weight'].shape == (256, 4, 3, 3)
    assert sd['backbone.layer2.0.conv2.weight'].shape == (512, 8, 3, 3)
    assert sd['backbone.layer2.0.conv2.conv_offset.weight'].shape == (18, 512, 3, 3)
    assert sd['backbone.layer3.22.conv2.weight'].shape == (1024, 16, 3, 3)
    assert sd['backbone.layer4.2.conv2.weight'].shape == (2048, 32, 3, 3)
    assert sd['backbone.layer4.2.conv3.weight'].shape == (2048, 2048, 1, 1)
    model = model.to(dev).train()
    tr = Trainer(model, lr=0.01)
    data = synthetic_batch(2, 256, 320, 311, device=dev, seed=1)
    for _ in range(2):
        out = tr.train_step(data)
    assert torch.isfinite(out['loss'].detach()).item()
    assert torch.isfinite(tr.flat.flat).all().item()
