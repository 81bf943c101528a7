"""GPU twin of tests/test_oracle_ops.py: the HIP operators (through the C ABI) against the reference-held known
answers (build/lib/mmdet/ops/nms/nms_wrapper.py:25-34,80-88), the gradcheck recipe of
build/lib/mmdet/ops/roi_align/gradcheck.py:9-29 and closed forms that do not depend on the C oracle at all."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_oracle_ops import NMS_KAT, SOFT_NMS_KAT, _gradcheck_inputs

pytestmark = pytest.mark.gpu
CL = torch.channels_last


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda:0')


def test_nms_known_answer_of_the_reference_wrapper(dev):
    from htd_amd import mmcv_ops as M
    d = torch.from_numpy(NMS_KAT).to(dev)
    dets, inds = M.nms(d[:, :4].contiguous(), d[:, 4].contiguous(), 0.6)
    assert len(inds) == len(dets) == 3
    assert inds.tolist() == [0, 3, 4]
    assert torch.equal(dets, d[inds])
    # batched_nms with one class is the same call
    dets2, keep2 = M.batched_nms(d[:, :4].contiguous(), d[:, 4].contiguous(), torch.zeros(7, dtype=torch.long, device=dev),
                                 dict(type='nms', iou_threshold=0.6))
    assert keep2.tolist() == [0, 3, 4] and torch.equal(dets2, dets)


def test_soft_nms_known_answer_of_the_reference_wrapper(dev):
    from htd_amd.soft_nms import soft_nms
    d = torch.from_numpy(SOFT_NMS_KAT).to(dev)
    dets, inds = soft_nms(d[:, :4].contiguous(), d[:, 4].contiguous(), 0.6, sigma=0.5, min_score=1e-3, method='linear')
    assert len(inds) == len(dets) == 5
    assert sorted(inds.tolist()) == [0, 1, 2, 3, 4]
    assert len(soft_nms(d[:, :4].contiguous(), d[:, 4].contiguous(), 0.6, sigma=0.5, min_score=1e-3, method='linear',
                        offset=1)[1]) == 3


def _ra(feat, rois, out, scale, sr, aligned, dev):
    from htd_amd import mmcv_ops as M
    return M.roi_align(feat.to(dev).contiguous(memory_format=CL), rois.to(dev), out, scale, sr, 'avg', aligned).cpu()


def test_roi_align_constant_map_gives_the_constant(dev):
    feat = torch.full((2, 4, 20, 24), 2.5)
    rois = torch.tensor([[0, 3.3, 2.1, 17.9, 15.2], [1, 0.7, 0.9, 23.0, 19.0], [1, 10.0, 10.0, 10.5, 10.2]])
    for sr in (0, 2):
        out = _ra(feat, rois, 7, 1.0, sr, True, dev)
        torch.testing.assert_close(out, torch.full_like(out, 2.5), rtol=0, atol=2e-6)


@pytest.mark.parametrize('bin_px', [1, 2, 3])
def test_roi_align_integer_aligned_roi_is_crop_or_avg_pool(dev, bin_px):
    g = torch.Generator().manual_seed(bin_px)
    feat = torch.randn(2, 8, 40, 44, generator=g)
    x1, y1 = 6, 4
    rois = torch.tensor([[1, x1, y1, x1 + 7 * bin_px, y1 + 7 * bin_px]], dtype=torch.float32)
    want = F.avg_pool2d(feat[1:2, :, y1:y1 + 7 * bin_px, x1:x1 + 7 * bin_px], bin_px)
    for sr in (bin_px, 0):
        torch.testing.assert_close(_ra(feat, rois, 7, 1.0, sr, True, dev), want, rtol=1e-6, atol=1e-6)
    out = _ra(feat, rois * torch.tensor([1, 4, 4, 4, 4.]), 7, 0.25, 0, True, dev)
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-6)


def test_roi_align_affine_map_gives_the_value_at_the_bin_centre(dev):
    H, W = 48, 64
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing='ij')
    a, b, c = 0.37, -0.21, 1.5
    feat = (a * yy + b * xx + c).float()[None, None].repeat(2, 4, 1, 1)
    g = torch.Generator().manual_seed(1)
    scale = 0.25
    xy = torch.rand(40, 2, generator=g) * torch.tensor([W * 0.5, H * 0.5]) / scale + 4
    wh = torch.rand(40, 2, generator=g) * torch.tensor([W * 0.4, H * 0.4]) / scale + 1
    rois = torch.cat([torch.randint(0, 2, (40, 1), generator=g).float(), xy, xy + wh], 1)
    out = _ra(feat, rois, 7, scale, 0, True, dev)
    r = rois.double()
    sx, sy = r[:, 1] * scale - 0.5, r[:, 2] * scale - 0.5
    bw, bh = (r[:, 3] - r[:, 1]) * scale / 7, (r[:, 4] - r[:, 2]) * scale / 7
    k = torch.arange(7, dtype=torch.float64) + 0.5
    want = a * (sy[:, None] + k[None] * bh[:, None])[:, :, None] + b * (sx[:, None] + k[None] * bw[:, None])[:, None, :] + c
    for ch in range(4):
        torch.testing.assert_close(out[:, ch].double(), want, rtol=1e-5, atol=1e-4)


def test_roi_align_outside_the_map_and_legacy_mode(dev):
    feat = torch.ones(1, 4, 8, 8)
    far = torch.tensor([[0, 100., 100., 120., 120.]])
    assert _ra(feat, far, 7, 1.0, 0, True, dev).abs().sum() == 0
    tiny = torch.tensor([[0, 3., 3., 3., 3.]])
    yy, xx = torch.meshgrid(torch.arange(8.), torch.arange(8.), indexing='ij')
    ramp = (xx + 10 * yy)[None, None].repeat(1, 4, 1, 1)
    torch.testing.assert_close(_ra(ramp, tiny, 1, 1.0, 1, False, dev)[0, 0].view(()), torch.tensor(38.5))
    torch.testing.assert_close(_ra(ramp, tiny, 1, 1.0, 1, True, dev)[0, 0].view(()), torch.tensor(27.5))


@pytest.mark.parametrize('sampling_ratio', [0, 2])
def test_roi_align_backward_is_the_adjoint_and_passes_the_reference_gradcheck_recipe(dev, sampling_ratio):
    """Sizes, eps and atol of build/lib/mmdet/ops/roi_align/gradcheck.py:9-29; no oracle involved."""
    from htd_amd import mmcv_ops as M
    feat, rois, scale = _gradcheck_inputs()
    g = torch.randn(20, 16, 3, 3, generator=torch.Generator().manual_seed(1))
    f = feat.to(dev).contiguous(memory_format=CL).requires_grad_()
    out = M.roi_align(f, rois.to(dev), 3, scale, sampling_ratio, 'avg', True)
    out.backward(g.to(dev))
    gin = f.grad.cpu()
    lhs = (out.detach().cpu().double() * g.double()).sum()
    rhs = (feat.double() * gin.double()).sum()
    torch.testing.assert_close(lhs, rhs, rtol=1e-5, atol=1e-4)
    eps = 1e-3
    rng = np.random.RandomState(2)
    rd = rois.to(dev)
    for _ in range(24):
        b, c, y, x = rng.randint(2), rng.randint(16), rng.randint(15), rng.randint(15)
        fp, fm = feat.clone(), feat.clone()
        fp[b, c, y, x] += eps
        fm[b, c, y, x] -= eps
        op = M.roi_align(fp.to(dev).contiguous(memory_format=CL), rd, 3, scale, sampling_ratio, 'avg', True).cpu().double()
        om = M.roi_align(fm.to(dev).contiguous(memory_format=CL), rd, 3, scale, sampling_ratio, 'avg', True).cpu().double()
        num = ((op - om) * g.double()).sum() / (2 * eps)
        assert abs(float(num) - float(gin[b, c, y, x])) < 1e-3, (b, c, y, x, float(num), float(gin[b, c, y, x]))


# --------------------------------------------------------------------------- deformable convolution
def _dcn(x, off, w, stride, pad, dil, dev, mask=None):
    from htd_amd.dcn import deform_conv2d
    y = deform_conv2d(x.to(dev).contiguous(memory_format=CL), off.to(dev).contiguous(memory_format=CL),
                      w.to(dev).contiguous(memory_format=CL), stride, pad, dil,
                      mask=None if mask is None else mask.to(dev).contiguous(memory_format=CL))
    return y.cpu()


@pytest.mark.parametrize('stride,pad,dil', [(1, 1, 1), (2, 1, 1), (1, 2, 2), (1, 0, 1)])
def test_dcn_zero_offsets_is_plain_convolution(dev, stride, pad, dil):
    g = torch.Generator().manual_seed(stride + pad)
    x = torch.randn(2, 16, 11, 13, generator=g)
    w = torch.randn(8, 16, 3, 3, generator=g) / 12
    ref = F.conv2d(x.double(), w.double(), None, stride, pad, dil).float()
    off = torch.zeros(2, 18, ref.shape[2], ref.shape[3])
    torch.testing.assert_close(_dcn(x, off, w, stride, pad, dil, dev), ref, rtol=1e-5, atol=1e-5)
    m = torch.full((2, 9, ref.shape[2], ref.shape[3]), 0.25)
    torch.testing.assert_close(_dcn(x, off, w, stride, pad, dil, dev, mask=m), 0.25 * ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('dy,dx', [(1, 0), (0, -2), (-3, 2)])
def test_dcn_integer_offsets_is_convolution_read_further_along(dev, dy, dx):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 12, 14, generator=g)
    w = torch.randn(8, 16, 3, 3, generator=g) / 12
    H, W = x.shape[2:]
    full = F.conv2d(F.pad(x.double(), (4, 4, 4, 4)), w.double()).float()
    ref = full[:, :, 3 + dy:3 + dy + H, 3 + dx:3 + dx + W]
    off = torch.zeros(1, 9, 2, H, W)
    off[:, :, 0] = dy
    off[:, :, 1] = dx
    torch.testing.assert_close(_dcn(x, off.view(1, 18, H, W), w, 1, 1, 1, dev), ref, rtol=1e-5, atol=1e-5)


def test_dcn_gradients_against_fp64_finite_differences(dev):
    """goffset / gmask / gx / gw of the HIP kernels against central differences of an fp64 evaluation of the same
    definition (torch restatement, itself gradcheck'ed in tests/test_oracle_ops.py) -- directional derivatives along
    random directions, sampling positions kept away from the bilinear kinks."""
    from htd_amd.dcn import deform_conv2d
    from oracle import ops as O
    g = torch.Generator().manual_seed(9)
    B, C, H, W, Co = 2, 16, 9, 10, 8
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / 12
    off = torch.randn(B, 18, H, W, generator=g) * 1.3
    off = torch.floor(off) + (off - torch.floor(off)).clamp(0.15, 0.85)
    mask = torch.rand(B, 9, H, W, generator=g)
    go = torch.randn(B, Co, H, W, generator=g)
    xd = x.to(dev).contiguous(memory_format=CL).requires_grad_()
    od = off.to(dev).contiguous(memory_format=CL).requires_grad_()
    wd = w.to(dev).contiguous(memory_format=CL).requires_grad_()
    md = mask.to(dev).contiguous(memory_format=CL).requires_grad_()
    deform_conv2d(xd, od, wd, 1, 1, 1, mask=md).backward(go.to(dev))

    def f64(x_, o_, w_, m_):
        return (O.deform_conv2d_autograd(x_.double(), o_.double(), w_.double(), 1, 1, 1, mask=m_.double()) * go.double()).sum()
    eps = 1e-4
    for name, t, grad, idx in (('x', x, xd.grad, 0), ('offset', off, od.grad, 1), ('weight', w, wd.grad, 2), ('mask', mask, md.grad, 3)):
        for trial in range(3):
            d = torch.randn(t.shape, generator=g)
            args_p = [x, off, w, mask]
            args_m = [x, off, w, mask]
            args_p[idx] = t.double() + eps * d.double()
            args_m[idx] = t.double() - eps * d.double()
            num = float((f64(*args_p) - f64(*args_m)) / (2 * eps))
            ana = float((grad.cpu().double() * d.double()).sum())
            assert abs(num - ana) <= 2e-4 * max(1.0, abs(num)), (name, trial, num, ana)
