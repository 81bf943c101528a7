"""Pins for the CPU restatement of the mmcv-full 1.2.1 operators (oracle/c/htd_oracle_ops.c, oracle/ops.py).

mmcv's sources are absent from /root/reference, so these operators cannot be compared with the real thing.  What the
reference tree does hold about them is used here:
  * the two known-answer examples of the stale wrapper docstrings
    (build/lib/mmdet/ops/nms/nms_wrapper.py:25-34: 7 dets, thr 0.6 => 3 kept; :80-88: soft-NMS => 5 kept);
  * the gradcheck recipe of build/lib/mmdet/ops/roi_align/gradcheck.py:9-29 (sizes, atol / eps).
Everything else is a closed form that any correct implementation of the documented semantics must satisfy (constant
and affine maps, integer-aligned RoIs == crop / avg_pool2d, adjoint identity, zero / integer offsets == plain /
shifted convolution, fp64 finite differences).  tests/test_gpu_op_pins.py runs the same checks on the HIP kernels.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as O

# --------------------------------------------------------------------------- reference-held known answers
NMS_KAT = np.array([[49.1, 32.4, 51.0, 35.9, 0.9],
                    [49.3, 32.9, 51.0, 35.3, 0.9],
                    [49.2, 31.8, 51.0, 35.4, 0.5],
                    [35.1, 11.5, 39.1, 15.7, 0.5],
                    [35.6, 11.8, 39.3, 14.2, 0.5],
                    [35.3, 11.5, 39.9, 14.5, 0.4],
                    [35.2, 11.7, 39.7, 15.7, 0.3]], dtype=np.float32)       # nms_wrapper.py:25-31
SOFT_NMS_KAT = np.array([[4., 3., 5., 3., 0.9],
                         [4., 3., 5., 4., 0.9],
                         [3., 1., 3., 1., 0.5],
                         [3., 1., 3., 1., 0.5],
                         [3., 1., 3., 1., 0.4],
                         [3., 1., 3., 1., 0.0]], dtype=np.float32)          # nms_wrapper.py:80-85


def test_nms_known_answer_of_the_reference_wrapper():
    d = torch.from_numpy(NMS_KAT)
    dets, inds = O.nms(d[:, :4], d[:, 4], 0.6)
    assert len(inds) == len(dets) == 3                                        # nms_wrapper.py:32-34
    # which three (IoUs by hand: 0-1 0.614, 0-2 0.699, 3-4 0.486, 3-5 0.594, 4-5 0.643, 3-6 0.813)
    assert inds.tolist() == [0, 3, 4]
    assert torch.equal(dets, d[inds])


def test_soft_nms_known_answer_of_the_reference_wrapper():
    d = torch.from_numpy(SOFT_NMS_KAT)
    dets, inds = O.soft_nms(d[:, :4], d[:, 4], 0.6, sigma=0.5, min_score=1e-3, method='linear')
    assert len(inds) == len(dets) == 5                                        # nms_wrapper.py:86-88
    assert sorted(inds.tolist()) == [0, 1, 2, 3, 4]                           # only the score-0.0 box falls under min_score
    # the example separates the two box conventions: with the legacy "+1" widths the degenerate boxes overlap fully
    # and linear decay removes two more
    assert len(O.soft_nms(d[:, :4], d[:, 4], 0.6, sigma=0.5, min_score=1e-3, method='linear', offset=1)[1]) == 3


def test_nms_is_greedy_suppression_in_score_order():
    """Independent O(n^2) torch formulation on random clustered boxes, incl. a score tie and a duplicate box."""
    g = torch.Generator().manual_seed(0)
    n = 300
    c = torch.rand(n // 6, 2, generator=g) * 200
    ctr = c[torch.randint(0, n // 6, (n, ), generator=g)] + torch.randn(n, 2, generator=g) * 4
    wh = torch.rand(n, 2, generator=g) * 40 + 8
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores = torch.rand(n, generator=g)
    scores[5] = scores[3]
    boxes[7] = boxes[2]
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt, rb = torch.max(b[:, None, :2], b[None, :, :2]), torch.min(b[:, None, 2:], b[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    iou = inter / (area[:, None] + area[None] - inter)
    alive = torch.ones(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if alive[i]:
            keep.append(int(order[i]))
            alive &= ~(iou[i] > 0.5) | (torch.arange(n) <= i)
    assert O.nms(boxes, scores, 0.5)[1].tolist() == keep


# --------------------------------------------------------------------------- RoIAlign closed forms
def test_roi_align_constant_map_gives_the_constant():
    feat = torch.full((2, 3, 20, 24), 2.5)
    rois = torch.tensor([[0, 3.3, 2.1, 17.9, 15.2], [1, 0.7, 0.9, 23.0, 19.0], [1, 10.0, 10.0, 10.5, 10.2]])
    for sr in (0, 2):
        out = O.roi_align_fwd(feat, rois, 7, 1.0, sr, True)
        torch.testing.assert_close(out, torch.full_like(out, 2.5), rtol=0, atol=1e-6)


@pytest.mark.parametrize('bin_px', [1, 2, 3])
def test_roi_align_integer_aligned_roi_is_crop_or_avg_pool(bin_px):
    """aligned=True shifts by -0.5, so an RoI whose bins are bin_px whole pixels wide samples pixel centres exactly:
    sampling_ratio = bin_px (and the adaptive grid ceil(bin) = bin_px) => avg_pool2d(crop, bin_px)."""
    g = torch.Generator().manual_seed(bin_px)
    feat = torch.randn(2, 5, 40, 44, generator=g)
    x1, y1 = 6, 4
    rois = torch.tensor([[1, x1, y1, x1 + 7 * bin_px, y1 + 7 * bin_px]], dtype=torch.float32)
    crop = feat[1:2, :, y1:y1 + 7 * bin_px, x1:x1 + 7 * bin_px]
    want = F.avg_pool2d(crop, bin_px)
    for sr in (bin_px, 0):
        out = O.roi_align_fwd(feat, rois, 7, 1.0, sr, True)
        torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-6)
    # the same box given in image coordinates of a stride-4 level
    out = O.roi_align_fwd(feat, rois * torch.tensor([1, 4, 4, 4, 4.]), 7, 0.25, 0, True)
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-6)


def test_roi_align_affine_map_gives_the_value_at_the_bin_centre():
    """Bilinear interpolation reproduces an affine map and a symmetric sample grid averages to the bin centre: for
    RoIs whose samples stay inside [0, H-1] x [0, W-1], out[i, j] = f(centre of bin (i, j)) in the -0.5-shifted
    frame -- checks the shift, the scale, the bin geometry and the adaptive grid together."""
    H, W = 48, 64
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing='ij')
    a, b, c = 0.37, -0.21, 1.5
    feat = (a * yy + b * xx + c).float()[None, None].repeat(2, 1, 1, 1)
    g = torch.Generator().manual_seed(1)
    scale = 0.25
    xy = torch.rand(40, 2, generator=g) * torch.tensor([W * 0.5, H * 0.5]) / scale + 4
    wh = torch.rand(40, 2, generator=g) * torch.tensor([W * 0.4, H * 0.4]) / scale + 1
    rois = torch.cat([torch.randint(0, 2, (40, 1), generator=g).float(), xy, xy + wh], 1)
    out = O.roi_align_fwd(feat, rois, 7, scale, 0, True)
    r = rois.double()
    sx, sy = r[:, 1] * scale - 0.5, r[:, 2] * scale - 0.5
    bw, bh = (r[:, 3] - r[:, 1]) * scale / 7, (r[:, 4] - r[:, 2]) * scale / 7
    k = torch.arange(7, dtype=torch.float64) + 0.5
    cy = sy[:, None] + k[None] * bh[:, None]
    cx = sx[:, None] + k[None] * bw[:, None]
    want = a * cy[:, :, None] + b * cx[:, None, :] + c
    torch.testing.assert_close(out[:, 0].double(), want, rtol=1e-5, atol=1e-4)


def test_roi_align_outside_the_map_and_legacy_mode():
    feat = torch.ones(1, 1, 8, 8)
    far = torch.tensor([[0, 100., 100., 120., 120.]])
    assert O.roi_align_fwd(feat, far, 7, 1.0, 0, True).abs().sum() == 0        # samples beyond [-1, H]: zero
    # aligned=False: no shift and RoI sizes are floored at one pixel (roi_align.py:97-118 of the stale wrapper)
    tiny = torch.tensor([[0, 3., 3., 3., 3.]])
    yy, xx = torch.meshgrid(torch.arange(8.), torch.arange(8.), indexing='ij')
    ramp = (xx + 10 * yy)[None, None]
    out = O.roi_align_fwd(ramp, tiny, 1, 1.0, 1, False)                        # one bin, one sample at (3.5, 3.5)
    torch.testing.assert_close(out.view(()), torch.tensor(3.5 + 35.0))
    out = O.roi_align_fwd(ramp, tiny, 1, 1.0, 1, True)                         # aligned: zero-size box, sample at 2.5
    torch.testing.assert_close(out.view(()), torch.tensor(2.5 + 25.0))


def _gradcheck_inputs(seed=0):
    """The configuration of build/lib/mmdet/ops/roi_align/gradcheck.py:9-24."""
    rng = np.random.RandomState(seed)
    feat_size, spatial_scale, num_imgs, num_rois = 15, 1.0 / 8, 2, 20
    img_size = feat_size / spatial_scale
    batch_ind = rng.randint(num_imgs, size=(num_rois, 1))
    rois = rng.rand(num_rois, 4) * img_size * 0.5
    rois[:, 2:] += img_size * 0.5
    rois = torch.from_numpy(np.hstack((batch_ind, rois))).float()
    feat = torch.randn(num_imgs, 16, feat_size, feat_size, generator=torch.Generator().manual_seed(seed))
    return feat, rois, spatial_scale


@pytest.mark.parametrize('sampling_ratio', [0, 2])
def test_roi_align_backward_is_the_adjoint_of_forward(sampling_ratio):
    """RoIAlign is linear in the features, so <fwd(x), g> = <x, bwd(g)> exactly (up to fp32 summation), and the
    finite-difference Jacobian of gradcheck.py equals fwd applied to unit maps."""
    feat, rois, scale = _gradcheck_inputs()
    g = torch.randn(20, 16, 3, 3, generator=torch.Generator().manual_seed(1))
    out = O.roi_align_fwd(feat, rois, 3, scale, sampling_ratio, True)
    gin = O.roi_align_bwd(g, rois, feat.shape, scale, sampling_ratio, True)
    lhs = (out.double() * g.double()).sum()
    rhs = (feat.double() * gin.double()).sum()
    torch.testing.assert_close(lhs, rhs, rtol=1e-5, atol=1e-4)
    # finite differences, eps / atol of the reference recipe (gradcheck.py:27-29), on a sample of input elements
    eps = 1e-3
    rng = np.random.RandomState(2)
    for _ in range(24):
        b, c, y, x = rng.randint(2), rng.randint(16), rng.randint(15), rng.randint(15)
        fp = feat.clone()
        fp[b, c, y, x] += eps
        fm = feat.clone()
        fm[b, c, y, x] -= eps
        num = ((O.roi_align_fwd(fp, rois, 3, scale, sampling_ratio, True).double() -
                O.roi_align_fwd(fm, rois, 3, scale, sampling_ratio, True).double()) * g.double()).sum() / (2 * eps)
        assert abs(float(num) - float(gin[b, c, y, x])) < 1e-3, (b, c, y, x, float(num), float(gin[b, c, y, x]))


def test_roi_align_gradcheck_recipe_of_the_reference():
    """torch.autograd.gradcheck exactly as gradcheck.py:27-29 calls it (fp32 inputs, atol = eps = 1e-3)."""
    feat, rois, scale = _gradcheck_inputs(3)
    feat = feat[:, :2].clone().requires_grad_()             # 2 channels keep the dense Jacobian small
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                      # "input is not double precision" -- as in the reference
        assert torch.autograd.gradcheck(O.RoIAlign(3, scale), (feat, rois), atol=1e-3, eps=1e-3)
        assert torch.autograd.gradcheck(O.RoIAlign(3, scale, 2), (feat, rois), atol=1e-3, eps=1e-3)


# --------------------------------------------------------------------------- deformable convolution closed forms
@pytest.mark.parametrize('stride,pad,dil', [(1, 1, 1), (2, 1, 1), (1, 2, 2), (1, 0, 1)])
def test_dcn_zero_offsets_is_plain_convolution(stride, pad, dil):
    g = torch.Generator().manual_seed(stride + pad)
    x = torch.randn(2, 6, 11, 13, generator=g)
    w = torch.randn(4, 6, 3, 3, generator=g)
    ref = F.conv2d(x, w, None, stride, pad, dil)
    off = torch.zeros(2, 18, ref.shape[2], ref.shape[3])
    torch.testing.assert_close(O.deform_conv2d(x, off, w, stride, pad, dil), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(O.deform_conv2d_autograd(x, off, w, stride, pad, dil), ref, rtol=1e-5, atol=1e-5)
    # v2 with a constant mask scales the output
    m = torch.full((2, 9, ref.shape[2], ref.shape[3]), 0.25)
    torch.testing.assert_close(O.deform_conv2d(x, off, w, stride, pad, dil, mask=m), 0.25 * ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('dy,dx', [(1, 0), (0, -2), (-3, 2)])
def test_dcn_integer_offsets_is_convolution_of_the_shifted_input(dy, dx):
    """Every tap displaced by the same whole (dy, dx) samples x at p + k + (dy, dx): the plain convolution read
    (dy, dx) further along (x taken as zero outside the map).  Also fixes the channel order of the offset tensor ([dy0, dx0, dy1, dx1, ...],
    build/lib/mmdet/ops/dcn/deform_conv.py:261-267)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 12, 14, generator=g)
    w = torch.randn(3, 4, 3, 3, generator=g)
    H, W = x.shape[2:]
    # output p reads x (zero outside) at p - 1 + k + d: a 'valid' convolution over x padded by 4, cropped at 3 + d
    full = F.conv2d(F.pad(x, (4, 4, 4, 4)), w)
    ref = full[:, :, 3 + dy:3 + dy + H, 3 + dx:3 + dx + W]
    off = torch.zeros(1, 9, 2, H, W)
    off[:, :, 0] = dy
    off[:, :, 1] = dx
    off = off.view(1, 18, H, W)
    torch.testing.assert_close(O.deform_conv2d(x, off, w, 1, 1, 1), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(O.deform_conv2d_autograd(x, off, w, 1, 1, 1), ref, rtol=1e-5, atol=1e-5)


def test_dcn_single_tap_offset_moves_only_that_tap():
    """A 1-hot kernel (only tap (0, 2)) with a fractional offset on that tap: the output is the bilinear sample."""
    x = torch.arange(30, dtype=torch.float32).view(1, 1, 5, 6)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 0, 2] = 1.0
    off = torch.zeros(1, 9, 2, 5, 6)
    off[:, 2, 0] = 0.5          # dy of tap 2
    off[:, 2, 1] = -0.25        # dx of tap 2
    out = O.deform_conv2d(x, off.view(1, 18, 5, 6), w, 1, 1, 1)
    # output (2, 2) samples x at (2 - 1 + 0 + 0.5, 2 - 1 + 2 - 0.25) = (1.5, 2.75); x = 6 y + x is affine
    torch.testing.assert_close(out[0, 0, 2, 2], torch.tensor(6 * 1.5 + 2.75))


def test_dcn_two_restatements_agree_and_gradients_pass_fp64_finite_differences():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 3, 6, 7, generator=g, dtype=torch.float64)
    w = torch.randn(2, 3, 3, 3, generator=g, dtype=torch.float64)
    off = torch.randn(1, 18, 6, 7, generator=g, dtype=torch.float64) * 1.3
    # keep sampling positions away from whole pixels, where the bilinear kernel has kinks
    frac = off - torch.floor(off)
    off = torch.floor(off) + frac.clamp(0.1, 0.9)
    mask = torch.rand(1, 9, 6, 7, generator=g, dtype=torch.float64)
    c = O.deform_conv2d(x.float(), off.float(), w.float(), 1, 1, 1, mask=mask.float())
    t = O.deform_conv2d_autograd(x, off, w, 1, 1, 1, mask=mask)
    torch.testing.assert_close(c.double(), t, rtol=1e-4, atol=1e-4)
    x.requires_grad_(), off.requires_grad_(), w.requires_grad_(), mask.requires_grad_()
    assert torch.autograd.gradcheck(lambda a, o, ww, m: O.deform_conv2d_autograd(a, o, ww, 1, 1, 1, mask=m),
                                    (x, off, w, mask), eps=1e-6, atol=1e-5)
    # stride 2, no mask (the v1 'DCN' the HTD config uses)
    off2 = off.detach()[:, :, :3, :4].clone().requires_grad_()
    assert torch.autograd.gradcheck(lambda a, o, ww: O.deform_conv2d_autograd(a, o, ww, 2, 1, 1), (x, off2, w),
                                    eps=1e-6, atol=1e-5)
