"""Parity of the HIP operators (through the C ABI) against the CPU oracle, on a real MI355X.
Bit-exact for index work (NMS keep sets / order), fp32 tolerance stated per test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda:0')


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def rand_rois(gen, n, B, H, W, stride, big=False):
    img_w, img_h = W * stride, H * stride
    size = torch.rand(n, generator=gen) * (0.9 if big else 0.3) * min(img_w, img_h) + 2
    ar = torch.exp(torch.rand(n, generator=gen) - 0.5)
    w, h = size * ar, size / ar
    cx, cy = torch.rand(n, generator=gen) * img_w, torch.rand(n, generator=gen) * img_h
    b = torch.randint(0, B, (n, ), generator=gen).float()
    rois = torch.stack([b, (cx - w / 2).clamp(0, img_w), (cy - h / 2).clamp(0, img_h), (cx + w / 2).clamp(0, img_w),
                        (cy + h / 2).clamp(0, img_h)], 1)
    return rois


@pytest.mark.parametrize('C,H,W,stride,n,big', [(256, 40, 56, 4, 64, False), (256, 20, 28, 8, 33, True),
                                                (8, 9, 7, 16, 20, True), (512, 12, 12, 32, 5, True),
                                                # >= 256 RoIs: the row-wise backward kernel (one atomic per pixel)
                                                (256, 40, 56, 4, 300, False), (64, 20, 28, 8, 400, True)])
def test_roi_align_fwd_bwd(dev, C, H, W, stride, n, big):
    from htd_amd import mmcv_ops as M
    from oracle import ops as O
    gen = torch.Generator().manual_seed(C + n)
    feat = torch.randn(2, C, H, W, generator=gen)
    rois = rand_rois(gen, n, 2, H, W, stride, big)
    rois[0, 1:] = torch.tensor([0., 0., 0., 0.])                      # degenerate: zero-size box
    rois[1, 1:] = torch.tensor([0., 0., W * stride, H * stride])      # whole image
    ref = O.roi_align_fwd(feat, rois, 7, 1.0 / stride, 0, True)
    f = cl(feat.to(dev)).requires_grad_()
    out = M.roi_align(f, rois.to(dev), 7, 1.0 / stride, 0, 'avg', True)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)   # fp32, separable-weight summation order
    go = torch.randn(out.shape, generator=gen)
    out.backward(go.to(dev))
    gref = O.roi_align_bwd(go, rois, feat.shape, 1.0 / stride, 0, True)
    torch.testing.assert_close(f.grad.cpu(), gref, rtol=1e-4, atol=1e-4)
    # aligned=False and fixed sampling_ratio
    ref2 = O.roi_align_fwd(feat, rois, 7, 1.0 / stride, 2, False)
    out2 = M.roi_align(f.detach(), rois.to(dev), 7, 1.0 / stride, 2, 'avg', False)
    torch.testing.assert_close(out2.cpu(), ref2, rtol=1e-5, atol=2e-5)


def test_roi_align_rejects_cpu_and_bad_rois(dev):
    from htd_amd import mmcv_ops as M
    with pytest.raises(NotImplementedError):
        M.roi_align(torch.zeros(1, 4, 4, 4), torch.zeros(1, 5), 7)
    with pytest.raises(AssertionError):
        M.roi_align(torch.zeros(1, 4, 4, 4, device=dev), torch.zeros(1, 4, device=dev), 7)
    with pytest.raises(ValueError):  # C not a multiple of 4: C-ABI argument error -> ValueError
        M.roi_align(torch.zeros(1, 3, 4, 4, device=dev), torch.zeros(1, 5, device=dev), 7)
    out = M.roi_align(torch.zeros(1, 4, 4, 4, device=dev), torch.zeros(0, 5, device=dev), 7)   # empty input
    assert out.shape == (0, 4, 7, 7)


def test_roi_align_levels_matches_per_level(dev):
    from htd_amd import mmcv_ops as M
    from oracle import boxes as B, detector as D
    gen = torch.Generator().manual_seed(3)
    feats = [torch.randn(2, 256, 64 // s, 96 // s, generator=gen) for s in (1, 2, 4, 8)]
    rois = rand_rois(gen, 150, 2, 64, 96, 4, big=True)
    ref = D.single_roi_extract(feats, rois)
    lv = B.map_roi_levels(rois, 4)
    fd = [cl(f.to(dev)).requires_grad_() for f in feats]
    out = M.roi_align_levels(fd, rois.to(dev), lv.to(dev), 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)
    go = torch.randn(out.shape, generator=gen)
    out.backward(go.to(dev))
    fr = [f.clone().requires_grad_() for f in feats]
    (D.single_roi_extract(fr, rois) * go).sum().backward()
    for a, b in zip(fd, fr):
        ref_grad = b.grad if b.grad is not None else torch.zeros_like(b)   # level without RoIs
        torch.testing.assert_close(a.grad.cpu(), ref_grad, rtol=1e-4, atol=1e-4)


def test_roi_align_levels_backward_one_launch_equals_per_level(dev):
    """htd_roi_align_levels_bwd_gather (all levels in one grid, coarsest first) against one htd_roi_align_bwd_gather launch per
    level: the same strips sum the same RoIs in the same order, so the maps are equal BIT FOR BIT -- fresh maps, maps that
    accumulate into a handed-down gradient (chain=True), a level without RoIs, a level that needs no gradient."""
    from htd_amd import mmcv_ops as M
    from oracle import boxes as B
    gen = torch.Generator().manual_seed(5)
    feats = [torch.randn(3, 64, 96 // s, 136 // s, generator=gen) for s in (1, 2, 4, 8)]
    rois = rand_rois(gen, 400, 3, 96, 136, 4, big=True)
    lv = B.map_roi_levels(rois, 4)
    lv[lv == 1] = 2                                             # level 1 stays empty
    go = torch.randn(400, 64, 7, 7, generator=gen).to(dev)
    res = {}
    try:
        for one in (True, False):
            M.ROI_BWD_ONE_LAUNCH = one
            fd = [cl(f.to(dev)).requires_grad_(i != 3) for i, f in enumerate(feats)]      # the coarsest level needs no gradient
            taps = M.PyramidTaps(fd)
            a = M.roi_align_levels(taps, rois.to(dev), lv.to(dev), 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
            b = M.roi_align_levels(taps, rois.to(dev).flip(0), lv.to(dev).flip(0), 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32])   # second consumer
            ((a * go).sum() + (b * go.flip(0) * 0.5).sum()).backward()
            res[one] = [f.grad for f in fd]
    finally:
        M.ROI_BWD_ONE_LAUNCH = True
    assert res[True][3] is None and res[False][3] is None
    for x, y in zip(res[True][:3], res[False][:3]):
        assert x is not None and torch.equal(x, y)
    assert float(res[True][1].abs().max()) == 0.0               # the empty level: zeros, written not accumulated


def test_plain_and_fused_with_row_stash_matches_the_tensor_formulation(dev):
    """mmcv_ops.plain_and_fused(+ RowStash) / select_rows_via against cat([x, x + g[img]]) and index_select: values, and the
    gradients of x and g when (a) both outputs are used, (b) no row is selected (a batch without positives), (c) the batch
    output is unused (only the selected rows feed the loss)."""
    from htd_amd import mmcv_ops as M
    gen = torch.Generator().manual_seed(9)
    n, C, B = 37, 64, 3
    x0 = torch.randn(n, C, 7, 7, generator=gen)
    g0 = torch.randn(B, C, 1, 1, generator=gen)
    img = torch.sort(torch.randint(0, B, (n, ), generator=gen))[0]
    rois = torch.cat([img[:, None].float(), torch.rand(n, 4, generator=gen) * 50], 1)
    rows = torch.tensor([0, 3, 4, 11, 36])
    wb = torch.randn(2 * n, C, 7, 7, generator=gen)
    wr = torch.randn(len(rows), C, 7, 7, generator=gen)
    for case in ('both', 'no rows', 'rows only'):
        xr, gr = x0.clone().requires_grad_(), g0.clone().requires_grad_()
        both_r = torch.cat([xr, xr + gr[img]], 0)
        loss_r = (both_r * wb).sum() if case != 'rows only' else xr.sum() * 0
        if case != 'no rows':
            loss_r = loss_r + (torch.index_select(xr, 0, rows) * wr).sum()
        loss_r.backward()
        xd, gd = cl(x0.to(dev)).requires_grad_(), g0.to(dev).requires_grad_()
        stash = M.RowStash()
        both = M.plain_and_fused(xd, rois.to(dev), gd, stash)
        assert torch.equal(both.cpu(), both_r.detach())
        loss = (both * wb.to(dev)).sum() if case != 'rows only' else xd.sum() * 0
        if case != 'no rows':
            sel = M.select_rows_via(stash, rows.to(dev))
            assert torch.equal(sel.cpu(), x0[rows])
            loss = loss + (sel * wr.to(dev)).sum()
        loss.backward()
        torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-6, atol=1e-6)
        if case != 'rows only':
            torch.testing.assert_close(gd.grad.cpu(), gr.grad, rtol=1e-5, atol=1e-4)
        assert stash.grad is None and not stash.pending            # handed over and cleared


def test_roi_align_levels_backward_gather_without_rois(dev):
    """htd_roi_align_levels_bwd_gather with n == 0: fresh maps are zero-filled, accumulated maps keep their contents."""
    import ctypes
    from htd_amd import capi
    L, B, C = 3, 2, 8
    maps = [torch.full((B, C, 16 >> i, 24 >> i), 3.0, device=dev).contiguous(memory_format=torch.channels_last) for i in range(L)]
    ptrs = (ctypes.c_void_p * L)(*[m.data_ptr() for m in maps])
    Hs, Ws = (ctypes.c_int * L)(16, 8, 4), (ctypes.c_int * L)(24, 12, 6)
    sc, ac = (ctypes.c_float * L)(0.25, 0.125, 0.0625), (ctypes.c_int * L)(0, 1, 0)
    capi.call('htd_roi_align_levels_bwd_gather', None, None, None, ptrs, Hs, Ws, sc, ac, L, 0, B, C, 7, 7, 0, 1, None,
              capi.current_stream_ptr())
    torch.cuda.synchronize()
    assert float(maps[0].abs().max()) == 0.0 and float(maps[2].abs().max()) == 0.0
    assert float((maps[1] - 3.0).abs().max()) == 0.0


def clustered_boxes(gen, n, span=300.):
    k = max(1, n // 6)
    centers = torch.rand(k, 2, generator=gen) * span
    c = centers[torch.randint(0, k, (n, ), generator=gen)] + torch.randn(n, 2, generator=gen) * 4
    wh = torch.rand(n, 2, generator=gen) * 40 + 8
    return torch.cat([c - wh / 2, c + wh / 2], 1)


@pytest.mark.parametrize('n', [1, 63, 64, 65, 500, 3000])
@pytest.mark.parametrize('thr', [0.5, 0.7])
def test_nms_bit_exact(dev, n, thr):
    from htd_amd import mmcv_ops as M
    from oracle import ops as O
    gen = torch.Generator().manual_seed(n)
    boxes = clustered_boxes(gen, n)
    scores = torch.rand(n, generator=gen)
    if n > 10:
        scores[5] = scores[3]          # exact score tie: lower index first
        boxes[7] = boxes[2]            # duplicate box
    dets_r, keep_r = O.nms(boxes, scores, thr)
    dets, keep = M.nms(boxes.to(dev), scores.to(dev), thr)
    assert torch.equal(keep.cpu(), keep_r)
    assert torch.equal(dets.cpu(), dets_r)
    _, keep1 = M.nms(boxes.to(dev), scores.to(dev), thr, offset=1)
    assert torch.equal(keep1.cpu(), O.nms(boxes, scores, thr, 1)[1])


def test_nms_empty(dev):
    from htd_amd import mmcv_ops as M
    dets, keep = M.nms(torch.zeros(0, 4, device=dev), torch.zeros(0, device=dev), 0.5)
    assert dets.shape == (0, 5) and keep.numel() == 0


@pytest.mark.parametrize('n,ncls', [(800, 5), (5000, 80), (12000, 80)])
def test_batched_nms_bit_exact(dev, n, ncls):
    """incl. n >= 10000 where mmcv switches to its per-class loop (same keep set and order)."""
    from htd_amd import mmcv_ops as M
    from oracle import ops as O
    gen = torch.Generator().manual_seed(n)
    boxes = clustered_boxes(gen, n, span=800.)
    scores = torch.rand(n, generator=gen)
    idxs = torch.randint(0, ncls, (n, ), generator=gen)
    cfg = dict(type='nms', iou_threshold=0.5)
    dets_r, keep_r = O.batched_nms(boxes, scores, idxs, cfg)
    dets, keep = M.batched_nms(boxes.to(dev), scores.to(dev), idxs.to(dev), cfg)
    assert torch.equal(keep.cpu(), keep_r)
    assert torch.equal(dets.cpu(), dets_r)


def test_fuse_global(dev):
    from htd_amd import mmcv_ops as M
    from oracle import detector as D
    gen = torch.Generator().manual_seed(0)
    n, C = 37, 256
    x = torch.randn(n, C, 7, 7, generator=gen)
    e = torch.randn(n, C, 7, 7, generator=gen)
    g = torch.randn(3, C, 1, 1, generator=gen)
    rois = torch.cat([torch.sort(torch.randint(0, 3, (n, 1), generator=gen).float(), 0)[0], torch.rand(n, 4, generator=gen)], 1)
    xr, er, gr = x.clone().requires_grad_(), e.clone().requires_grad_(), g.clone().requires_grad_()
    ref = D.fuse_global(xr, gr, rois) + 0.5 * er
    xd, ed, gd = cl(x.to(dev)).requires_grad_(), cl(e.to(dev)).requires_grad_(), g.to(dev).requires_grad_()
    out = M.fuse_global(xd, rois.to(dev), gd, ed, 0.5)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)
    go = torch.randn(out.shape, generator=gen)
    ref.backward(go)
    out.backward(go.to(dev))
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ed.grad.cpu(), er.grad, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(gd.grad.cpu(), gr.grad, rtol=1e-4, atol=1e-4)
    out2 = M.fuse_global(xd.detach(), rois.to(dev), gd.detach())
    torch.testing.assert_close(out2.cpu(), D.fuse_global(x, g, rois), rtol=1e-6, atol=1e-6)


def test_ba_fuse(dev):
    from htd_amd import mmcv_ops as M
    gen = torch.Generator().manual_seed(1)
    n, C, L = 19, 256, 4
    lv = [torch.randn(n, C, 7, 7, generator=gen) for _ in range(L)]
    att = torch.randn(L, n, generator=gen)
    lr = [t.clone().requires_grad_() for t in lv]
    ar = att.clone().requires_grad_()
    w = ar.softmax(0)
    mask = torch.ones(7, 7)
    mask[1:-1, 1:-1] = 0
    ref = sum(w[l].view(n, 1, 1, 1) * lr[l] for l in range(L)) + lr[0] * mask   # adaptative_roi_extractor.py:80-91
    ld = [cl(t.to(dev)).requires_grad_() for t in lv]
    ad = att.to(dev).requires_grad_()
    out = M.ba_fuse(ad, ld, 1)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    go = torch.randn(out.shape, generator=gen)
    ref.backward(go)
    out.backward(go.to(dev))
    for a, b in zip(ld, lr):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ad.grad.cpu(), ar.grad, rtol=1e-4, atol=1e-4)


def test_global_avg_pool_and_group_norm(dev):
    from htd_amd import mmcv_ops as M
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(21, 576, 7, 7, generator=gen)
    xr = x.clone().requires_grad_()
    xd = cl(x.to(dev)).requires_grad_()
    ref = F.adaptive_avg_pool2d(xr, 1)
    out = M.global_avg_pool(xd)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)
    go = torch.randn(ref.shape, generator=gen)
    ref.backward(go)
    out.backward(go.to(dev))
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-6, atol=1e-7)
    # GN36 + ReLU (htd_bbox_head.py:48,89,111)
    gamma, beta = torch.randn(576, generator=gen), torch.randn(576, generator=gen)
    for relu in (True, False):
        xr = x.clone().requires_grad_()
        gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
        ref = F.group_norm(xr, 36, gr, br, 1e-5)
        ref = F.relu(ref) if relu else ref
        xd = cl(x.to(dev)).requires_grad_()
        gd, bd = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
        out = M.group_norm_relu(xd, gd, bd, 36, 1e-5, relu)
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)
        go = torch.randn(ref.shape, generator=gen)
        ref.backward(go)
        out.backward(go.to(dev))
        torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(gd.grad.cpu(), gr.grad, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(bd.grad.cpu(), br.grad, rtol=1e-3, atol=1e-3)


def test_sgd_step(dev):
    from htd_amd import mmcv_ops as M
    gen = torch.Generator().manual_seed(4)
    n = 100003
    p, g = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    pr = p.clone().requires_grad_()
    opt = torch.optim.SGD([pr], lr=0.02, momentum=0.9, weight_decay=1e-4)
    pd, md = p.to(dev), torch.zeros(n, device=dev)
    lr = torch.tensor([0.02], device=dev)
    for _ in range(3):
        pr.grad = g.clone()
        opt.step()
        M.sgd_momentum_step_(pd, g.to(dev), md, lr, 0.9, 1e-4)
    torch.testing.assert_close(pd.cpu(), pr.detach(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('method', ['linear', 'naive', 'gaussian'])
def test_soft_nms_matches_oracle(dev, method):
    """Kept set, order and decayed scores of the device soft-NMS against the sequential C restatement."""
    from htd_amd.soft_nms import soft_nms, soft_nms_batched
    from oracle import ops as O
    gen = torch.Generator().manual_seed(7)
    boxes = clustered_boxes(gen, 400)
    scores = torch.rand(400, generator=gen)
    dets_r, inds_r = O.soft_nms(boxes, scores, 0.5, 0.5, 0.05, method)
    dets, inds = soft_nms(boxes.to(dev), scores.to(dev), 0.5, 0.5, 0.05, method)
    assert torch.equal(inds.cpu(), inds_r)
    tol = dict(rtol=0, atol=0) if method != 'gaussian' else dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dets.cpu(), dets_r, **tol)
    # batched over classes == mmcv's single call on offset-shifted boxes
    idxs = torch.randint(0, 7, (400, ), generator=gen)
    cfg = dict(type='soft_nms', iou_thr=0.5, min_score=0.05, method=method)
    dets_r, keep_r = O.batched_nms(boxes, scores, idxs, cfg)
    from htd_amd.mmcv_ops import batched_nms
    dets, keep = batched_nms(boxes.to(dev), scores.to(dev), idxs.to(dev), cfg)
    assert torch.equal(keep.cpu(), keep_r)
    torch.testing.assert_close(dets.cpu(), dets_r, **tol)


@pytest.mark.gpu
@pytest.mark.parametrize('shared,low_quality,K', [(True, True, 5), (False, False, 7), (False, True, 150), (True, True, 1)])
def test_max_iou_assign_kernel_matches_tensor_formulation(shared, low_quality, K):
    """htd_max_iou_assign against the tensor-op formulation of MaxIoUAssigner (itself pinned to the reference's
    known-answer tests on the CPU): identical assignment incl. ties (duplicate gts, boxes equal to gts),
    padded gts, invalid boxes, an image without gt."""
    import htd_amd.core.bbox as cb
    from htd_amd.core.bbox import MaxIoUAssigner
    torch.manual_seed(5)
    dev = torch.device('cuda:0')
    B, A = 3, 5000
    a = MaxIoUAssigner(pos_iou_thr=0.7 if low_quality else 0.5, neg_iou_thr=0.3 if low_quality else 0.5,
                       min_pos_iou=0.3 if low_quality else 0.5, match_low_quality=low_quality, ignore_iof_thr=-1)
    xy = torch.rand(B, A, 2) * 300
    wh = torch.rand(B, A, 2) * 120 + 2
    boxes = torch.cat([xy, xy + wh], -1)
    gxy = torch.rand(B, K, 2) * 300
    gwh = torch.rand(B, K, 2) * 150 + 4
    gts = torch.cat([gxy, gxy + gwh], -1)
    if K > 2:
        gts[:, 1] = gts[:, 0]                        # duplicate gt: argmax / low-quality ties
    boxes[:, :K] = gts if not shared else gts[:1]    # boxes equal to gts: IoU exactly 1
    gt_valid = torch.ones(B, K, dtype=torch.bool)
    gt_valid[1, K // 2:] = False                     # padded gts
    gt_valid[2] = False                              # image without gt
    gts = gts * gt_valid[..., None]
    box_valid = torch.rand(B, A) > 0.1
    if shared:
        boxes = boxes[0]
    boxes, gts, gt_valid, box_valid = boxes.to(dev), gts.to(dev), gt_valid.to(dev), box_valid.to(dev)
    got, got_ov = cb.batched_max_iou_assign(a, boxes, box_valid, gts, gt_valid)
    saved = cb._max_iou_assign_device
    try:
        cb._max_iou_assign_device = None             # force the tensor formulation
        a2 = MaxIoUAssigner(pos_iou_thr=a.pos_iou_thr, neg_iou_thr=a.neg_iou_thr, min_pos_iou=a.min_pos_iou,
                            match_low_quality=low_quality, ignore_iof_thr=-1)
        ref, ref_ov = cb._batched_max_iou_assign_tensor(a2, boxes, box_valid, gts, gt_valid)
    finally:
        cb._max_iou_assign_device = saved
    assert torch.equal(got, ref)
    assert torch.equal(got_ov, ref_ov)
    assert int((got > 0).sum()) > 0 and int((got == 0).sum()) > 0


@pytest.mark.parametrize('k,s,p,H,W', [(3, 2, 1, 37, 52), (2, 2, 0, 16, 24), (3, 1, 1, 9, 11)])
def test_max_pool2d_matches_aten(dev, k, s, p, H, W):
    """The stem's max pooling on NHWC maps: values and gradient routing (first maximum of the window, ties included)."""
    from htd_amd import mmcv_ops as M
    gen = torch.Generator().manual_seed(k * 10 + s)
    x = (torch.randn(2, 8, H, W, generator=gen) * 2).round() / 2          # quantised: many ties inside windows
    a = cl(x.to(dev)).requires_grad_()
    b = x.clone().requires_grad_()
    ya = M.max_pool2d(a, k, s, p)
    yb = torch.nn.functional.max_pool2d(b, k, s, p)
    assert torch.equal(ya.cpu(), yb)
    g = torch.randn(yb.shape, generator=gen)
    ya.backward(g.to(dev))
    yb.backward(g)
    torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=0, atol=1e-6)


@pytest.mark.parametrize('C,H,W,stride,n', [(256, 40, 56, 4, 300), (64, 20, 28, 8, 400), (8, 9, 7, 16, 20), (512, 50, 84, 16, 700)])
def test_roi_align_backward_gather_form(dev, C, H, W, stride, n):
    """The gather-form backward (no atomics): equal to the scatter kernels and to the oracle within fp32 summation order,
    bit-identical from run to run, correct when it accumulates onto an existing map (chained consumers), and it writes
    every pixel (a map full of NaNs must come back clean when accumulate = 0)."""
    from htd_amd import capi
    from oracle import ops as O
    gen = torch.Generator().manual_seed(C + n)
    B = 2
    rois = rand_rois(gen, n, B, H, W, stride, big=True)
    rois[0, 1:] = torch.tensor([0., 0., 0., 0.])
    rois[1, 1:] = torch.tensor([0., 0., W * stride, H * stride])
    go = torch.randn(n, C, 7, 7, generator=gen)
    gref = O.roi_align_bwd(go, rois, (B, C, H, W), 1.0 / stride, 0, True)
    god, rd = cl(go.to(dev)), rois.to(dev)
    L = capi.lib()
    ws = torch.empty(L.htd_roi_align_bwd_gather_workspace_bytes(n), dtype=torch.uint8, device=dev)

    def gather(into, accumulate):
        capi.call('htd_roi_align_bwd_gather', capi.ptr(god), capi.ptr(rd), None, 0, capi.ptr(into), n, B, C, H, W, 7, 7,
                  1.0 / stride, 0, 1, accumulate, capi.ptr(ws), capi.current_stream_ptr())
        return into
    a = gather(torch.full((B, C, H, W), float('nan'), device=dev).contiguous(memory_format=torch.channels_last), 0)
    b = gather(torch.empty((B, C, H, W), device=dev).contiguous(memory_format=torch.channels_last), 0)
    assert torch.equal(a, b)                                            # bit-stable, every pixel written
    torch.testing.assert_close(a.cpu(), gref, rtol=1e-4, atol=1e-4)
    scat = torch.zeros((B, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    capi.call('htd_roi_align_bwd', capi.ptr(god), capi.ptr(rd), None, 0, capi.ptr(scat), n, B, C, H, W, 7, 7, 1.0 / stride, 0, 1,
              capi.current_stream_ptr())
    torch.testing.assert_close(a, scat, rtol=1e-4, atol=1e-4)
    base = torch.randn(B, C, H, W, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    c = gather(base.clone(memory_format=torch.channels_last), 1)
    torch.testing.assert_close(c, base + a, rtol=1e-6, atol=1e-6)
    # RoIs of another pyramid level are skipped
    lv = torch.zeros(n, dtype=torch.int64, device=dev)
    lv[::2] = 1
    d = torch.empty_like(a)
    capi.call('htd_roi_align_bwd_gather', capi.ptr(god), capi.ptr(rd), capi.ptr(lv), 1, capi.ptr(d), n, B, C, H, W, 7, 7,
              1.0 / stride, 0, 1, 0, capi.ptr(ws), capi.current_stream_ptr())
    gref1 = O.roi_align_bwd(go[::2], rois[::2], (B, C, H, W), 1.0 / stride, 0, True)
    torch.testing.assert_close(d.cpu(), gref1, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_segmented_topk_equals_stable_descending_sort(dev):
    """htd_segmented_topk == keys.sort(descending=True, stable=True)[:k] per segment, positions included (rpn_head.py:122-133):
    ties at the threshold (quantised keys), all-equal segments, segments shorter than k, one-element and empty picks,
    negative keys, the full P2 level size, and torch.sort's treatment of -0 / +0 (equal) and NaN (largest, either sign)."""
    import htd_amd.mmcv_ops as M
    g = torch.Generator().manual_seed(11)
    parts = [
        torch.rand(201600, generator=g),                                   # P2 level, k = 2000
        (torch.rand(50400, generator=g) * 64).floor() / 64,                # heavy ties (64 distinct values)
        torch.full((5000, ), 0.5),                                         # every key equal: the first k positions
        torch.rand(819, generator=g),                                      # shorter than k: everything, sorted
        torch.randn(12345, generator=g),                                   # negative keys
        torch.tensor([3.0]),                                               # one key
        torch.rand(4096 * 3, generator=g).round(decimals=2),               # chunk-aligned length, ties across chunks
        torch.rand(100, generator=g),                                      # k = 0
        torch.cat([-torch.zeros(10), torch.zeros(7000), torch.ones(3), -torch.zeros(10)]),   # -0 == +0: ties by position
        torch.cat([torch.rand(5000, generator=g) - 0.5, torch.tensor([float('nan'), -float('nan')]),      # NaNs of either sign
                   torch.tensor([float('inf'), -float('inf'), -0.0, 0.0])]),                             # sort first, like torch
    ]
    ks = [2000, 2000, 777, 819, 2048, 1, 1000, 0, 15, 2048]
    keys = torch.cat(parts)
    segs, off = [], 0
    for p, k in zip(parts, ks):
        segs.append((off, p.numel(), k))
        off += p.numel()
    idx, val = M.segmented_topk(keys.to(dev), segs)
    torch.cuda.synchronize()
    idx, val = idx.cpu(), val.cpu()
    o = 0
    for i, (p, k) in enumerate(zip(parts, ks)):
        rv, ri = p.sort(descending=True, stable=True)
        # bit-level equality: a -0 key comes back as -0, a NaN as the NaN it was (values are read back from the keys)
        assert torch.equal(val[o:o + k].view(torch.int32), rv[:k].view(torch.int32)), i
        assert torch.equal(idx[o:o + k], ri[:k]), i
        o += k
    assert o == idx.numel()


@pytest.mark.gpu
@pytest.mark.parametrize('B,A,num,frac,ub,pos_rate', [(4, 268569, 256, 0.5, -1, 2e-4), (2, 2008, 512, 0.25, -1, 0.05),
                                                      (3, 5000, 64, 0.5, 3, 0.001), (2, 300, 512, 0.25, -1, 0.5),
                                                      (1, 9000, 128, 0.5, -1, 0.0)])
def test_random_sample_kernel_equals_tensor_formulation(dev, B, A, num, frac, ub, pos_rate):
    """htd_random_sample == the sort-based formulation of RandomSampler (base_sampler.py:34-101): same masks, counts and slot
    order, with duplicate keys (ties by index), fewer candidates than slots, the neg_pos_ub bound and no positives at all."""
    from htd_amd.core import bbox as Bx
    g = torch.Generator().manual_seed(A + num)
    r = torch.rand(B, A, generator=g)
    assigned = torch.where(r < pos_rate, torch.randint(1, 9, (B, A), generator=g), torch.zeros(B, A, dtype=torch.long))
    assigned = torch.where(torch.rand(B, A, generator=g) < 0.1, torch.full_like(assigned, -1), assigned)
    keys = (torch.rand(B, A, generator=g) * 4096).floor() / 4096              # duplicates: ties go to the lower index
    pos_ref, neg_ref = Bx.batched_random_sample(assigned, num, frac, ub, keys)  # CPU: the tensor formulation
    slots = num
    pos, neg, counts, order = Bx.random_sample_device(assigned.to(dev), keys.to(dev), num, frac, ub, slots=slots)
    torch.cuda.synchronize()
    assert torch.equal(pos.cpu(), pos_ref) and torch.equal(neg.cpu(), neg_ref)
    assert counts[:, 0].cpu().tolist() == pos_ref.sum(1).tolist() and counts[:, 1].cpu().tolist() == neg_ref.sum(1).tolist()
    ar = torch.arange(A).expand(B, A)
    ref_order = torch.where(pos_ref, ar, torch.where(neg_ref, ar + A, ar + 2 * A)).argsort(dim=1, stable=True)[:, :slots]
    for b in range(B):
        n = int(pos_ref[b].sum() + neg_ref[b].sum())
        assert order[b, :n].cpu().tolist() == ref_order[b, :n].tolist()
        assert int(order[b, n:].abs().sum()) == 0


@pytest.mark.gpu
def test_map_roi_levels_kernel_equals_the_tensor_formula():
    """htd_map_roi_levels against the reference's five tensor operations (single_level_roi_extractor.py:32-51), including boxes
    whose scale sits exactly on a level boundary (112, 224, 448 px) and empty slots"""
    from htd_amd.detector.roi_extractors import map_roi_levels
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(4)
    n = 5000
    size = torch.exp(torch.rand(n, generator=g) * 6.0 + 1.0)
    ar = torch.exp(torch.rand(n, generator=g) * 2 - 1)
    x1, y1 = torch.rand(n, generator=g) * 900, torch.rand(n, generator=g) * 600
    rois = torch.stack([torch.zeros(n), x1, y1, x1 + size * ar, y1 + size / ar], 1)
    edge = torch.tensor([[0, 10., 20., 10 + s, 20 + s] for s in (56., 111.99999, 112., 112.00001, 224., 448., 447.99997, 896., 0., 1e-3)])
    rois = torch.cat([rois, edge, torch.zeros(7, 5)]).to(dev)
    for L in (4, 5, 1):
        scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
        ref = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(min=0, max=L - 1).long()
        got = map_roi_levels(rois, L, 56)
        assert got.dtype == torch.int64 and torch.equal(got, ref)


@pytest.mark.gpu
def test_rpn_flat_heads_equal_the_per_level_formulation():
    """RPNHead.forward hands loss and proposal stage the flat per-anchor tensors made by ONE gather launch
    (htd_rpn_heads_gather / _scatter): losses, their gradients and the proposals must equal the per-level split / reshape /
    concatenate formulation (anchor_head.py:172-269, rpn_head.py:78-168) bit for bit -- the same values in the same order."""
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.detector import rpn_head as R
    from htd_amd.runner import synthetic_batch
    dev = torch.device('cuda:0')
    cfg = htd_config(50)
    cfg.train_cfg.rpn_proposal.update(nms_pre=300, nms_post=200, max_num=200)
    torch.manual_seed(3)
    det = build_htd_detector(cfg=cfg).to(dev).train()
    data = synthetic_batch(2, 192, 256, 250, device=dev, seed=5)
    feats = [f.detach() for f in det.extract_feat(data['img'])]
    out = {}
    saved = R.RPN_FLAT_HEADS
    try:
        for flat in (False, True):
            R.RPN_FLAT_HEADS = flat
            det.zero_grad()
            fs = [f.clone().requires_grad_() for f in feats]
            torch.manual_seed(7)
            cls, reg = det.rpn_head(fs)
            assert (getattr(cls, 'flat', None) is not None) == flat
            losses = det.rpn_head.loss(cls, reg, data['gt_bboxes'], data['img_metas'])
            props = det.rpn_head.get_bboxes(cls, reg, data['img_metas'], cfg=det.train_cfg.rpn_proposal, padded=True)
            total = sum(v for vs in losses.values() for v in vs)
            total.backward()
            out[flat] = ([c.detach().clone() for c in cls], [r.detach().clone() for r in reg], float(total), props,
                         [f.grad.clone() for f in fs], {n: p.grad.clone() for n, p in det.rpn_head.named_parameters()})
    finally:
        R.RPN_FLAT_HEADS = saved
    a, b = out[False], out[True]
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert x.shape == y.shape and torch.equal(x, y)
    assert a[2] == b[2]
    assert torch.equal(a[3][0], b[3][0]) and torch.equal(a[3][1], b[3][1])
    for x, y in zip(a[4], b[4]):
        assert torch.equal(x, y)
    for n in a[5]:
        assert torch.equal(a[5][n], b[5][n]), n


@pytest.mark.gpu
@pytest.mark.parametrize('add_gt,P,S', [(True, 300, 64), (False, 300, 64), (True, 20, 64), (True, 1000, 512)])
def test_static_samples_finish_kernel_equals_the_tensor_formulation(add_gt, P, S):
    """core.bbox.static_assign_and_sample: the fixed-slot result written by htd_static_samples_finish against the gather / compare
    / concatenate formulation it replaces (sampling_result.py:40-60 per image) -- every member torch.equal, including images
    without gt, fewer candidates than slots (P = 20) and unused slots."""
    from htd_amd.core import bbox as Bx
    from htd_amd.core.bbox import MaxIoUAssigner, RandomSampler, set_sample_keys
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(P + S)
    B = 3
    props = torch.rand(B, P, 4, generator=g) * 200
    props[..., 2:] = props[..., :2] + torch.rand(B, P, 2, generator=g) * 120 + 1
    pvalid = torch.rand(B, P, generator=g) > 0.1
    props = (props * pvalid[..., None]).to(dev)
    gts = [torch.tensor([[20., 30., 120., 160.], [100., 50., 220., 180.], [5., 5., 60., 40.]]).to(dev), torch.zeros(0, 4).to(dev),
           torch.tensor([[60., 60., 190., 200.]]).to(dev)]
    labels = [torch.tensor([3, 7, 1]).to(dev), torch.zeros(0, dtype=torch.int64).to(dev), torch.tensor([5]).to(dev)]
    assigner = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)
    sampler = RandomSampler(num=S, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=add_gt)
    coef = torch.tensor([12.9898, 78.233, 37.719, 93.989], device=dev)
    set_sample_keys(lambda cand: torch.frac(torch.sin((cand * coef).sum(-1)) * 43758.5453).abs())
    saved = Bx.STATIC_FINISH
    out = {}
    try:
        for fin in (False, True):
            Bx.STATIC_FINISH = fin
            out[fin] = Bx.static_assign_and_sample(assigner, sampler, props, pvalid.to(dev), gts, labels)
    finally:
        Bx.STATIC_FINISH = saved
        set_sample_keys(None)
    a, b = out[False], out[True]
    for name in ('boxes', 'valid', 'is_pos', 'npos', 'nneg', 'pos_gt_bboxes', 'pos_gt_labels', 'pos_is_gt'):
        x, y = getattr(a, name), getattr(b, name)
        assert x.shape == y.shape and x.dtype == y.dtype and torch.equal(x, y), name
    assert int(b.npos.sum()) > 0 and bool((b.valid.sum(1) <= S).all())


@pytest.mark.parametrize('N,n,shape', [(2048, 512, (256, 7, 7)), (96, 16, (64, 7, 7)), (33, 33, (8, 1, 1)), (10, 0, (16, 3, 3))])
def test_row_selection_and_its_adjoint(N, n, shape):
    """htd_rows_gather / htd_rows_add (the stage-2 positives' rows of the RoI tiles, htd_roi_head.py:163-166) against
    torch.index_select / index_add_, bit for bit; and through mmcv_ops.select_rows_via the gradient of the selected rows joins
    the PlainAndFused node's sum exactly as the dense autograd formulation's."""
    from htd_amd import capi, mmcv_ops as M
    dev = torch.device('cuda:0')
    CL = torch.channels_last
    g = torch.Generator().manual_seed(N + n)
    x = torch.randn(N, *shape, generator=g).to(dev).contiguous(memory_format=CL)
    rows = torch.randperm(N, generator=g)[:n].to(dev)
    F = x[0].numel()
    out = torch.empty((n, ) + shape, device=dev).contiguous(memory_format=CL)
    capi.call('htd_rows_gather', capi.ptr(x), capi.ptr(rows), capi.ptr(out), n, N, F, capi.current_stream_ptr())
    assert torch.equal(out, torch.index_select(x, 0, rows))
    upd = torch.randn(n, *shape, generator=g).to(dev).contiguous(memory_format=CL)
    gx = x.clone(memory_format=torch.preserve_format)
    capi.call('htd_rows_add', capi.ptr(upd), capi.ptr(rows), capi.ptr(gx), n, N, F, capi.current_stream_ptr())
    assert torch.equal(gx, x.clone().index_add_(0, rows, upd))
    if n == 0 or shape[0] % 4:
        return
    B = 2
    rois = torch.cat([torch.randint(0, B, (N, 1), generator=g).float(), torch.rand(N, 4, generator=g) * 50], 1).to(dev)
    glob = torch.randn(B, shape[0], 1, 1, generator=g).to(dev)
    w = torch.randn(n, *shape, generator=g).to(dev)
    grads = []
    for via in (True, False):
        xr = x.clone(memory_format=torch.preserve_format).requires_grad_()
        if via:
            stash = M.RowStash()
            both = M.plain_and_fused(xr, rois, glob, stash)
            sel = M.select_rows_via(stash, rows)
        else:
            both = M.plain_and_fused(xr, rois, glob)
            sel = torch.index_select(xr, 0, rows)
        (both.square().sum() + (sel * w).sum()).backward()
        grads.append(xr.grad.clone())
    assert torch.equal(grads[0], grads[1])
