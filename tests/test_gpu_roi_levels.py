"""Every RoI pooled from every level in one launch each way (mmcv_ops.roi_align_all_levels, csrc/roi_align.hip): what
AdptRoIExtractor.forward (roi_extractors/adaptative_roi_extractor.py:66-76) does with one RoIAlign call per level.  Forward:
the bits of the per-level calls.  Backward (gather form, the RoIs of a strip dealt to the wavefronts of a group): the per-level
gradient within fp32 summation-order noise, the adjoint identity, and bit-identical from run to run."""
import pytest
import torch

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _setup(n, B, seed, clustered):
    from htd_amd import mmcv_ops as M
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(seed)
    H, W = 96, 160
    feats = [torch.randn(B, 256, H // s, W // s, generator=g).to(dev).contiguous(memory_format=CL) for s in (1, 2, 4, 8)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    img = torch.sort(torch.randint(0, B, (n, ), generator=g).float())[0]
    if clustered:       # jittered copies of a few boxes per image: every strip inside a box is reached by dozens of RoIs
        gts = torch.rand(B, 3, 4, generator=g)
        pick = gts[img.long(), torch.randint(0, 3, (n, ), generator=g)]
        cx, cy = pick[:, 0] * 4 * W, pick[:, 1] * 4 * H
        w, h = 40 + pick[:, 2] * 300, 40 + pick[:, 3] * 200
        jit = (torch.rand(n, 4, generator=g) - 0.5) * 12
        rois = torch.stack([img, cx - w / 2 + jit[:, 0], cy - h / 2 + jit[:, 1], cx + w / 2 + jit[:, 2], cy + h / 2 + jit[:, 3]], 1)
    else:
        size = torch.exp(torch.rand(n, generator=g) * 3.5 + 2.0)
        cx, cy = torch.rand(n, generator=g) * 4 * W, torch.rand(n, generator=g) * 4 * H
        rois = torch.stack([img, cx - size / 2, cy - size / 3, cx + size / 2, cy + size / 3], 1)
    rois[:, 1::2] = rois[:, 1::2].clamp(-20, 4 * W + 20)        # a little outside the image too
    rois[:, 2::2] = rois[:, 2::2].clamp(-20, 4 * H + 20)
    return M, feats, scales, rois.to(dev)


@pytest.mark.parametrize('n,B,clustered', [(3, 1, False), (40, 2, True), (600, 3, False), (512, 4, True)])
def test_all_levels_roi_align_matches_the_per_level_calls(n, B, clustered):
    M, feats, scales, rois = _setup(n, B, n + B, clustered)
    fa = [f.clone().requires_grad_() for f in feats]
    fb = [f.clone().requires_grad_() for f in feats]
    outs = M.roi_align_all_levels(fa, rois, 7, scales)
    refs = [M.roi_align(fb[i], rois, 7, scales[i], 0, 'avg', True) for i in range(4)]
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)
    g = torch.Generator().manual_seed(1)
    gos = [torch.randn(o.shape, generator=g).to(o.device).contiguous(memory_format=CL) for o in outs]
    torch.autograd.backward(outs, gos)
    torch.autograd.backward(refs, gos)
    for a, b in zip(fa, fb):
        scale = float(b.grad.abs().max())
        torch.testing.assert_close(a.grad, b.grad, rtol=2e-5, atol=2e-6 * scale)
    # adjoint identity <RoIAlign_l(f), g> = <f, RoIAlign_l^T(g)> per level, in fp64 sums
    for f, a, o, go in zip(feats, fa, outs, gos):
        lhs = float((o.detach().double() * go.double()).sum())
        rhs = float((f.double() * a.grad.double()).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    # one level unused: its map gets no gradient, the others are unchanged
    fc = [f.clone().requires_grad_() for f in feats]
    outs = M.roi_align_all_levels(fc, rois, 7, scales)
    torch.autograd.backward([outs[0], outs[2], outs[3]], [gos[0], gos[2], gos[3]])
    assert fc[1].grad is None
    for i in (0, 2, 3):
        assert torch.equal(fc[i].grad, fa[i].grad)          # same kernel, same order: bit-identical from run to run


def test_all_levels_backward_adds_into_chained_maps():
    """PyramidTaps: the gradient maps handed down the tap chain are accumulated into (accumulate = 1), not overwritten"""
    M, feats, scales, rois = _setup(64, 2, 9, True)
    base = [f.clone().requires_grad_() for f in feats]
    taps = M.PyramidTaps(base)
    outs = M.roi_align_all_levels(taps, rois, 7, scales)
    lv = torch.zeros(rois.size(0), dtype=torch.int64, device=rois.device) + 1
    single = M.roi_align_levels(taps, rois, lv, 7, scales)      # a second consumer further down the chain (level 1 only)
    loss = sum((o * o).sum() for o in outs) + single.sum()
    loss.backward()
    ref = [f.clone().requires_grad_() for f in feats]
    outs2 = [M.roi_align(ref[i], rois, 7, scales[i], 0, 'avg', True) for i in range(4)]
    single2 = M.roi_align(ref[1], rois, 7, scales[1], 0, 'avg', True)
    (sum((o * o).sum() for o in outs2) + single2.sum()).backward()
    for a, b in zip(base, ref):
        torch.testing.assert_close(a.grad, b.grad, rtol=2e-5, atol=2e-6 * float(b.grad.abs().max()))


@pytest.mark.parametrize('n,B,clustered', [(64, 1, False), (130, 2, True), (600, 3, False), (512, 4, True)])
def test_folded_backward_is_bit_identical_to_the_unfolded_one(n, B, clustered, monkeypatch):
    """htd_roi_align_{all_,}levels_bwd_gather_folded: the bins folded along y once per (RoI, map row) by roi_fold_kernel, the strips
    fetching folded vectors -- the same sums in the same order as the strips folding for themselves, so the same bits."""
    M, feats, scales, rois = _setup(n, B, 100 + n, clustered)
    from htd_amd.detector.roi_extractors import map_roi_levels
    lv = map_roi_levels(rois, 4)
    g = torch.Generator().manual_seed(2)

    def run(fold):
        monkeypatch.setattr(M, 'ROI_FOLD', fold)
        monkeypatch.setattr(M, 'ROI_FOLD_SINGLE', fold)
        fa = [f.clone().requires_grad_() for f in feats]
        outs = M.roi_align_all_levels(fa, rois, 7, scales)
        g.manual_seed(2)
        gos = [torch.randn(o.shape, generator=g).to(o.device).contiguous(memory_format=CL) for o in outs]
        torch.autograd.backward(outs, gos)
        fb = [f.clone().requires_grad_() for f in feats]
        one = M.roi_align_levels(fb, rois, lv, 7, scales)
        one.backward(gos[0])
        return [f.grad for f in fa], [f.grad for f in fb]

    a_all, a_one = run(True)
    b_all, b_one = run(False)
    for x, y in zip(a_all + a_one, b_all + b_one):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y)


def test_fold_is_skipped_past_the_workspace_cap(monkeypatch):
    M, feats, scales, rois = _setup(128, 2, 5, False)
    monkeypatch.setattr(M, 'ROI_FOLD_MAX_BYTES', 1 << 20)
    import ctypes
    Hs = (ctypes.c_int * 4)(*[f.shape[2] for f in feats])
    assert M._fold_workspace(128, Hs, 4, 7, 256, rois.device) is None
    monkeypatch.setattr(M, 'ROI_FOLD_MAX_BYTES', 8 << 30)
    ws = M._fold_workspace(128, Hs, 4, 7, 256, rois.device)
    assert ws is not None and ws.numel() == 128 * sum(f.shape[2] for f in feats) * 7 * 256 * 4
