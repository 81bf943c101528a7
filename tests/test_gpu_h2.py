"""H2 arithmetic of the 3x3 layers (csrc/conv_x3.hip, h2_scale): every fp32 product through two block-scaled fp16 pieces per
operand, a0 b0 + a0 b1 + a1 b0.  Role in the reference: cuDNN behind the 3x3 convolutions of backbones/resnet.py:260-300,
necks/fpn.py:77-90, dense_heads/rpn_head.py:25-27 and the regression branch of htd_bbox_head.py:77-113.

Bars: error against an fp64 reference no larger than RMS_BOUND x the fp32-input MFMA kernel's on the same data (forward and data
gradient, wide-dynamic-range inputs); integer-exact where fp32 is; the pieces, the scales and the maximum themselves bit-exact
against their tensor formulation; and the ranges the block scaling has to survive: tiny and huge tensors, one outlier 2^20 above
everything else, all zeros, a NaN."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last
RMS_BOUND, MAX_BOUND = 1.5, 2.0


@pytest.fixture
def h2():
    from htd_amd import capi, dense
    L = capi.lib()
    if not L.htd_conv2d_x3h_supported(256, 256, 3, 3, 1, 1, 1):
        pytest.skip('H2 arithmetic switched off')
    yield L
    L.htd_conv2d_set_h2(1)
    L.htd_conv2d_set_math(1)
    dense.new_step()


def _run(dense, x, w, gy, bias=None):
    """Forward + data gradient with both inputs carrying their maxima (as tensors written by the package's epilogues do): small or
    light layers take H2 only then (dense.H2_ABSMAX_MIN_WORK / _MIN_ELEMS)."""
    dense.new_step()
    xr = x.clone().requires_grad_()
    gy = gy.clone()
    dense.tag_amax(xr, dense.absmax(xr))
    dense.tag_amax(gy, dense.absmax(gy))
    y = dense.conv2d(xr, w, bias, 1, 1, 1)
    y.backward(gy)
    return y.detach(), xr.grad


@pytest.mark.parametrize('Ci,Co,H,W,B', [(256, 256, 40, 56, 2), (48, 96, 33, 47, 2), (576, 576, 7, 7, 16), (64, 64, 50, 70, 1)])
def test_h2_is_as_accurate_as_the_fp32_matrix_instructions(h2, Ci, Co, H, W, B):
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(Ci + H)
    x = (torch.randn(B, Ci, H, W, generator=g) * torch.exp(torch.randn(B, Ci, H, W, generator=g) * 2)).to(dev).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * torch.exp(torch.randn(Co, Ci, 3, 3, generator=g)) / (Ci * 9) ** 0.5).to(dev)
    w = w.contiguous(memory_format=CL)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    scale = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
    gy = torch.randn(ref.shape, generator=g).to(dev).contiguous(memory_format=CL)
    gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 1, 1)
    gscale = torch.nn.grad.conv2d_input(x.shape, w.double().abs(), gy.double().abs(), 1, 1)
    err = {}
    for mode in ('native', 'h2'):
        h2.htd_conv2d_set_math(0 if mode == 'native' else 1)
        h2.htd_conv2d_set_h2(1 if mode == 'h2' else 0)
        calls = []
        orig = dense.capi.call

        def spy(name, *a, **k):
            calls.append(name)
            return orig(name, *a, **k)
        dense.capi.call = spy
        try:
            y, gx = _run(dense, x, w, gy)
        finally:
            dense.capi.call = orig
        assert ('htd_conv2d_fwd_x3h' in calls) == (mode == 'h2') and ('htd_conv2d_bwd_data_x3h' in calls) == (mode == 'h2')
        ef, eg = (y.double() - ref).abs() / scale, (gx.double() - gref).abs() / gscale
        err[mode] = (float(ef.max()), float(ef.pow(2).mean().sqrt()), float(eg.max()), float(eg.pow(2).mean().sqrt()))
    n, h = err['native'], err['h2']
    assert h[1] <= RMS_BOUND * n[1] and h[3] <= RMS_BOUND * n[3], (n, h)
    assert h[0] <= MAX_BOUND * n[0] and h[2] <= MAX_BOUND * n[2], (n, h)


def test_h2_is_exact_on_small_integers(h2):
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-40, 41, (2, 64, 19, 23), generator=g).float().to(dev).contiguous(memory_format=CL)
    w = torch.randint(-9, 10, (96, 64, 3, 3), generator=g).float().to(dev).contiguous(memory_format=CL)
    b = torch.randint(-5, 6, (96, ), generator=g).float().to(dev)
    gy = torch.randint(-7, 8, (2, 96, 19, 23), generator=g).float().to(dev).contiguous(memory_format=CL)
    y, gx = _run(dense, x, w, gy, b)
    assert torch.equal(y.double(), F.conv2d(x.double(), w.double(), b.double(), 1, 1))
    assert torch.equal(gx.double(), torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 1, 1))


def test_absmax_kernel(h2):
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    for n in (1, 3, 4, 5, 1023, 4096 + 2, 3_000_001):
        x = torch.randn(n, generator=g).to(dev)
        x[n // 2] = -77.5
        dense.new_step()
        assert float(dense.absmax(x)) == 77.5
    x = torch.zeros(1000, device=dev)
    assert float(dense.absmax(x)) == 0.0
    x[17] = float('nan')
    assert torch.isnan(dense.absmax(x)).item()
    x[17] = float('-inf')
    assert float(dense.absmax(x)) == float('inf')
    # slots of one step are independent
    dense.new_step()
    a, b = dense.absmax(torch.full((64, ), 2.0, device=dev)), dense.absmax(torch.full((64, ), 0.5, device=dev))
    assert float(a) == 2.0 and float(b) == 0.5


def test_h2_weight_image_is_the_exact_two_piece_split(h2):
    """htd_conv2d_x3h_planes: row scale = the power of two that puts the row's largest magnitude into [2^14, 2^15); pieces
    a0 = fp16(s w), a1 = fp16(s w - a0) (round to nearest even); both operand orientations; the 'many' launch writes the same."""
    from htd_amd import capi, dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    Co, Ci = 80, 48
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * torch.exp(torch.randn(Co, 1, 1, 1, generator=g) * 3)).to(dev).contiguous(memory_format=CL)
    w[7] = 0.0                                         # an all-zero output channel
    for tr in (0, 1):
        N, K = (Ci, Co) if tr else (Co, Ci)
        Np = (N + 127) // 128 * 128
        nbytes = h2.htd_conv2d_x3_planes_bytes(Co, 3, 3, Ci, tr)
        img = torch.zeros(nbytes // 4, device=dev, dtype=torch.int32)
        capi.call('htd_conv2d_x3h_planes', capi.ptr(w), capi.ptr(img), Co, 3, 3, Ci, tr, capi.current_stream_ptr())
        nvec = 9 * (K // 16) * 6 * Np
        scales = img[nvec * 4:].view(torch.float32)
        inv, fwd = scales[:Np], scales[Np:2 * Np]
        # rows of the operand: forward w[n][tap][k]; transposed w[k][8 - tap][n]
        wk = w.permute(0, 2, 3, 1).reshape(Co, 9, Ci)            # [co][tap][ci]
        rows = wk if not tr else wk.flip(1).permute(2, 1, 0)      # [n][tap][k]
        amax = rows.abs().amax(dim=(1, 2))
        e = torch.where(amax > 0, 14 - torch.floor(torch.log2(amax.double())), torch.full_like(amax, 126).double()).clamp(max=126)
        assert torch.equal(fwd[:N].double().cpu(), torch.ldexp(torch.ones(N, dtype=torch.float64), e.cpu().int()))
        assert torch.equal((fwd[:N] * inv[:N]), torch.ones(N, device=dev))
        s = rows * fwd[:N, None, None]
        a0 = s.half()
        a1 = (s - a0.float()).half()
        planes = img[:nvec * 4].view(torch.float16).view(9, K // 16, 6, Np, 8)
        for piece, t in ((0, a0), (1, a1)):
            got = planes[:, :, 2 * piece:2 * piece + 2, :N]                                   # [tap][cs][half][n][8]
            want = t.view(N, 9, K // 16, 2, 8).permute(1, 2, 3, 0, 4)
            assert torch.equal(got, want), (tr, piece)
        assert float(a0.float().abs().max()) < 32768.0
        # the same image out of the step's table launch
        dense.new_step()
        dense.planes_many([(w, bool(tr))])
        many = dense.x3_planes(w, bool(tr))
        assert torch.equal(many[:nvec * 4].view(9, K // 16, 6, Np, 4)[:, :, :4, :N], img[:nvec * 4].view(9, K // 16, 6, Np, 4)[:, :, :4, :N])
        assert torch.equal(many[nvec * 4:nvec * 4 + 2 * Np], img[nvec * 4:nvec * 4 + 2 * Np])
    dense.new_step()


@pytest.mark.parametrize('case', ['tiny', 'huge', 'outlier', 'zeros', 'nan'])
def test_h2_ranges(h2, case):
    """What the per-tensor scale has to survive.  tiny / huge: 2^-100 / 2^100 times ordinary data (the scale is exact, the result
    is the scaled result of the ordinary data bit for bit); outlier: one element 2^20 above the rest -- the rest loses relative
    precision to fp16's subnormal floor (absolute precision 2^-40 of the tensor maximum per element), and the outputs the outlier
    dominates show the 22 significant bits of a two-piece operand; zeros in, zeros out; a NaN in the input poisons the outputs it reaches, like fp32."""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 64, 24, 30, generator=g).to(dev).contiguous(memory_format=CL)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev).contiguous(memory_format=CL)
    gy = torch.randn(1, 64, 24, 30, generator=g).to(dev).contiguous(memory_format=CL)
    y0, g0 = _run(dense, x, w, gy)
    if case in ('tiny', 'huge'):
        f = 2.0 ** (-100 if case == 'tiny' else 100)
        y, gx = _run(dense, x * f, w, gy * f)
        assert torch.equal(y, y0 * f) and torch.equal(gx, g0 * f)
    elif case == 'outlier':
        x2 = x.clone()
        x2[0, 5, 7, 9] = 2.0 ** 20
        y, _ = _run(dense, x2, w, gy)
        ref = F.conv2d(x2.double(), w.double(), None, 1, 1)
        err = (y.double() - ref).abs()
        mag = F.conv2d(x2.double().abs(), w.double().abs(), None, 1, 1)
        # two fp16 pieces carry 22 significant bits (x exact here, w 2^-22, the dropped a1 b1 2^-22), and the 16 products of one
        # matrix instruction are aligned to their largest before they are added -- both show where ONE product dominates a sum:
        # measured 2^-20.8 of the magnitude, bound 2^-20 (fp32 itself: 2^-24 per product, K 2^-24 per sum)
        far = torch.ones_like(err, dtype=torch.bool)
        far[:, :, 6:9, 8:11] = False
        assert float((err / mag)[~far].max()) <= 2.0 ** -20
        # outputs the outlier does not reach: the ordinary elements sit 2^20 below the tensor maximum and keep an absolute
        # precision of 2^-40 of it (= 2^-20 each), summed over K = 576 products with random signs
        assert float(err[far].max()) <= 2.0 ** -20 * float(w.abs().pow(2).sum(dim=(1, 2, 3)).sqrt().max()) * 6 + 2e-5
    elif case == 'zeros':
        y, gx = _run(dense, torch.zeros_like(x), w, torch.zeros_like(gy))
        assert float(y.abs().max()) == 0.0 and float(gx.abs().max()) == 0.0
    else:
        x2 = x.clone()
        x2[0, 3, 4, 5] = float('nan')
        y, _ = _run(dense, x2, w, gy)
        assert torch.isnan(y[0, :, 3:6, 4:7]).all()


@pytest.mark.parametrize('Ci,Co,k,H,W,B', [(256, 256, 3, 40, 56, 2), (64, 64, 3, 50, 70, 2), (1024, 256, 1, 50, 84, 2), (128, 512, 1, 33, 47, 3),
                                           (64, 256, 1, 60, 80, 2), (576, 576, 3, 7, 7, 16)])
def test_h2_weight_gradient_is_as_accurate_as_the_fp32_matrix_instructions(h2, Ci, Co, k, H, W, B):
    """htd_conv2d_bwd_weight_h2 (conv_wgrad_x3d_kernel / conv_wgrad_x3hd_kernel with H2 = true): both operands are activations,
    each scaled from its own tensor maximum; wide-dynamic-range data; error against fp64 relative to the accumulated magnitude."""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(Ci + H + k)
    x = (torch.randn(B, Ci, H, W, generator=g) * torch.exp(torch.randn(B, Ci, H, W, generator=g) * 2)).to(dev).contiguous(memory_format=CL)
    gy = (torch.randn(B, Co, H, W, generator=g) * torch.exp(torch.randn(B, Co, H, W, generator=g) * 2) * 1e-3).to(dev).contiguous(memory_format=CL)
    w = torch.zeros(Co, Ci, k, k, device=dev).contiguous(memory_format=CL)
    if not h2.htd_conv2d_bwd_weight_h2_supported(B, H, W, Ci, Co, k, k, 1, k // 2, 1):
        pytest.skip('layer not on the H2 weight-gradient kernels')
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), 1, k // 2)
    mag = torch.nn.grad.conv2d_weight(x.double().abs(), w.shape, gy.double().abs(), 1, k // 2)
    bref = gy.double().sum(dim=(0, 2, 3))
    err = {}
    for mode in ('native', 'h2'):
        h2.htd_conv2d_set_math(0 if mode == 'native' else 1)
        h2.htd_conv2d_set_h2(1 if mode == 'h2' else 0)
        dense.new_step()
        if mode == 'h2':                              # (1x1 layers take the arithmetic only when both maxima are carried)
            dense.tag_amax(x, dense.absmax(x))
            dense.tag_amax(gy, dense.absmax(gy))
        calls = []
        orig = dense.capi.call

        def spy(name, *a, **kw):
            calls.append(name)
            return orig(name, *a, **kw)
        dense.capi.call = spy
        try:
            gw, gb = dense._wgrad_raw(x, gy, w, 1, k // 2, 1, bias=True, overlap=False)
        finally:
            dense.capi.call = orig
        assert ('htd_conv2d_bwd_weight_h2' in calls) == (mode == 'h2')
        e = (gw.double() - ref).abs() / mag
        err[mode] = (float(e.max()), float(e.pow(2).mean().sqrt()))
        torch.testing.assert_close(gb.double(), bref, rtol=1e-4, atol=1e-4 * float(bref.abs().max()))
    n, h = err['native'], err['h2']
    assert h[1] <= RMS_BOUND * n[1] and h[0] <= MAX_BOUND * n[0], (n, h)


def test_h2_weight_gradient_accumulates_and_is_exact_on_integers(h2):
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(8)
    x = torch.randint(-30, 31, (2, 64, 17, 21), generator=g).float().to(dev).contiguous(memory_format=CL)
    gy = torch.randint(-9, 10, (2, 128, 17, 21), generator=g).float().to(dev).contiguous(memory_format=CL)
    for k in (1, 3):
        w = torch.zeros(128, 64, k, k, device=dev).contiguous(memory_format=CL)
        ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), 1, k // 2)
        dense.new_step()
        dense.tag_amax(x, dense.absmax(x))
        dense.tag_amax(gy, dense.absmax(gy))
        gw, _ = dense._wgrad_raw(x, gy, w, 1, k // 2, 1, overlap=False)
        assert torch.equal(gw.double(), ref)
        # accumulate form through the C-ABI: gw += the gradient
        from htd_amd import capi
        acc = gw.clone()
        nbytes = h2.htd_conv2d_wgrad_workspace_bytes(2, 17, 21, 64, 128, k, k, 1, k // 2, 1)
        ws = torch.empty(nbytes // 4 + 1, device=dev)
        capi.call('htd_conv2d_bwd_weight_h2', capi.ptr(x), capi.ptr(gy), capi.ptr(dense.absmax(x)), capi.ptr(dense.absmax(gy)),
                  capi.ptr(acc), None, 2, 17, 21, 64, 128, k, k, 1, k // 2, 1, 1, capi.ptr(ws), capi.current_stream_ptr())
        assert torch.equal(acc.double(), 2 * ref)


def test_carried_maximum_contract(h2):
    """dense.tag_amax / carried_amax: the maximum is remembered on the tensor object and by address (with a weak reference to the
    storage), and honoured only while address, element count and torch's version counter are what they were and the storage is
    the one that was written -- views of the whole tensor keep it (linear()'s reshapes, autograd's view nodes), an in-place torch
    op voids it, a copy or a partial / strided view never had it, drop_amax removes it (for kernels that write behind torch's back),
    a new tensor at the freed address does not inherit it; and it survives Function.apply and save_for_backward."""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    dense.new_step()
    x = torch.randn(2, 64, 9, 11, device=dev).contiguous(memory_format=CL)
    w = torch.randn(64, 64, 3, 3, device=dev).contiguous(memory_format=CL) * 0.05
    dense.tag_amax(x, dense.absmax(x))
    y = dense._fwd_raw(x, w, None, None, 1, 1, 1, True)
    am = dense.carried_amax(y)                                # left by the epilogue
    assert am is not None and float(am) == float(y.abs().max())
    assert dense.carried_amax(y.clone()) is None
    for v in (y.view_as(y), y.permute(0, 2, 3, 1), y.permute(0, 2, 3, 1).reshape(2, -1), y.detach()):
        got = dense.carried_amax(v)                           # the same elements under another shape
        assert got is not None and got.data_ptr() == am.data_ptr()
    assert dense.carried_amax(y[:1]) is None and dense.carried_amax(y[:, :32]) is None        # a part of the elements
    flat = y.permute(0, 2, 3, 1).reshape(-1)
    assert dense.carried_amax(flat[::2]) is None and dense.carried_amax(flat[1:]) is None
    big = torch.empty(4 * y.numel(), device=dev)
    half = big[:y.numel()]
    dense.tag_amax(half, dense.absmax(half.zero_()))
    assert dense.carried_amax(big[:2 * y.numel():2]) is None  # same address and element count, other elements
    # the address handed to another tensor: the weak reference to the storage that was written has expired
    ptr, shape = y.data_ptr(), y.shape
    keep = am.clone()
    del y, v, got, flat
    z = torch.empty(shape, device=dev).contiguous(memory_format=CL)
    if z.data_ptr() == ptr:
        assert dense.carried_amax(z) is None
    monkey = dense.H2_VIEWS
    try:
        dense.H2_VIEWS = False
        y = dense._fwd_raw(x, w, None, None, 1, 1, 1, True)
        assert dense.carried_amax(y) is not None and dense.carried_amax(y.view_as(y)) is None      # tensor objects only
    finally:
        dense.H2_VIEWS = monkey
    y = dense._fwd_raw(x, w, None, None, 1, 1, 1, True)
    assert float(keep) == float(dense.carried_amax(y))
    y.mul_(2.0)                                               # torch's version counter moved: stale
    assert dense.carried_amax(y) is None
    assert dense.carried_amax(y.view_as(y)) is None
    y2 = dense._fwd_raw(x, w, None, None, 1, 1, 1, True)
    dense.drop_amax(y2)
    assert dense.carried_amax(y2) is None and dense.carried_amax(y2.view_as(y2)) is None

    class Pass(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            mid = dense._fwd_raw(t, w, None, None, 1, 1, 1, True)         # an intermediate, saved as it is
            out = dense._fwd_raw(mid, w, None, None, 1, 1, 1, False)
            ctx.save_for_backward(mid, out)
            return out

        @staticmethod
        def backward(ctx, g):
            mid, out = ctx.saved_tensors
            # (a saved OUTPUT is re-wrapped by autograd when it is unpacked -- a new object: found by address)
            Pass.seen = (dense.carried_amax(mid) is not None, dense.carried_amax(out) is not None)
            return g
    xr = x.clone().requires_grad_()
    dense.tag_amax(xr, dense.absmax(xr))
    out = Pass.apply(xr)
    assert dense.carried_amax(out) is not None
    out.sum().backward()
    assert Pass.seen == (True, True)
    dense.new_step()


def test_stale_maximum_is_caught_by_the_guard(h2, monkeypatch):
    """A maximum that is NOT the tensor's (a caller's bug) puts infinities into the output; with the guard on (HTD_H2_GUARD / check
    mode) dense.h2_check() finds the launch whose epilogue left a non-finite maximum from a finite input maximum and fails loudly."""
    from htd_amd import capi, dense
    dev = torch.device('cuda:0')
    monkeypatch.setattr(dense, 'H2_GUARD', True)
    monkeypatch.setattr(dense, 'H2_CHECK', False)          # (check mode would refuse the wrong tag before the launch)
    dense.new_step()
    x = torch.randn(1, 64, 16, 16, device=dev).contiguous(memory_format=CL)
    w = torch.randn(64, 64, 3, 3, device=dev).contiguous(memory_format=CL) * 0.05
    wrong = dense.absmax(x * 1e-3)                            # a thousand times too small
    dense.tag_amax(x, wrong)
    dense.h2_check()                                          # clean so far
    y = dense._fwd_raw(x, w, None, None, 1, 1, 1, False)
    assert not torch.isfinite(y).all()
    with pytest.raises(RuntimeError, match='stale'):
        dense.h2_check()
    dense.h2_check()                                          # the flag was cleared
    dense.new_step()


@pytest.mark.parametrize('rows,C,with_y,with_bias', [(1000, 256, True, True), (777, 1024, True, False), (513, 30, True, True),
                                                      (2048, 1024, False, True), (64, 81, False, True)])
def test_masked_gradient_leaves_its_maximum(h2, rows, C, with_y, with_bias):
    """htd_bias_grad_relu_mask_amax: the same gm / gbias as htd_bias_grad_relu_mask, and max |gm| bit for bit (both kernels: 16-byte
    and scalar columns); a NaN in the gradient comes out as NaN."""
    from htd_amd import capi
    dev = torch.device('cuda:0')
    torch.manual_seed(rows + C)
    g = torch.randn(rows, C, device=dev) * 3
    y = torch.randn(rows, C, device=dev) if with_y else None
    for nan in (False, True):
        if nan:
            g[rows // 2, C // 3] = float('nan')
            if with_y:
                y[rows // 2, C // 3] = 1.0
        out = []
        for name in ('htd_bias_grad_relu_mask', 'htd_bias_grad_relu_mask_amax'):
            gm = torch.empty_like(g) if with_y else None
            gb = torch.empty(C, device=dev) if with_bias else None
            ws = torch.empty(2048 * C, device=dev) if with_bias else None
            slot = torch.zeros(1, device=dev)
            extra = (capi.ptr(slot), ) if name.endswith('_amax') else ()
            capi.call(name, capi.ptr(g), capi.ptr(y), capi.ptr(gm), capi.ptr(gb), rows, C, capi.ptr(ws), *extra, capi.current_stream_ptr())
            out.append((gm, gb, slot))
        (gm0, gb0, _), (gm1, gb1, slot) = out
        if with_y:
            assert torch.equal(gm0.nan_to_num(7.0), gm1.nan_to_num(7.0))
        if with_bias:
            assert torch.equal(gb0.nan_to_num(7.0), gb1.nan_to_num(7.0))
        ref = (gm1 if with_y else g).abs().max()
        assert torch.isnan(slot).item() if nan else float(slot) == float(ref)


def test_roi_align_levels_leaves_its_maximum(h2):
    """htd_roi_align_levels_fwd_amax: the tiles of htd_roi_align_levels_fwd and max |out| bit for bit; through mmcv_ops the tag reaches
    the FC layer behind the flatten (a view) and that layer runs on H2 -- forward and weight gradient."""
    from htd_amd import capi, dense, mmcv_ops
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    feats = [torch.randn(2, 64, 80 >> l, 96 >> l, device=dev).contiguous(memory_format=CL).requires_grad_() for l in range(4)]
    n = 300
    xy = torch.rand(n, 2, device=dev) * torch.tensor([300.0, 250.0], device=dev)
    wh = torch.rand(n, 2, device=dev) * 120 + 4
    rois = torch.cat([torch.randint(0, 2, (n, 1), device=dev).float(), xy, xy + wh], 1)
    lv = torch.randint(0, 4, (n, ), device=dev)
    scales = (0.25, 0.125, 0.0625, 0.03125)
    h2.htd_conv2d_set_h2(0)
    ref = mmcv_ops._RoIAlignLevels.apply(rois, lv, 7, scales, 2, True, False, *[f.detach() for f in feats])
    assert dense.carried_amax(ref) is None
    h2.htd_conv2d_set_h2(1)
    dense.new_step()
    out = mmcv_ops._RoIAlignLevels.apply(rois, lv, 7, scales, 2, True, False, *feats)
    assert torch.equal(out, ref)
    am = dense.carried_amax(out)
    assert am is not None and float(am) == float(out.detach().abs().max())
    fc = torch.nn.Linear(64 * 49, 256).to(dev)
    xf = out.permute(0, 2, 3, 1).reshape(n, -1)
    assert dense.carried_amax(xf) is not None
    capi.profile_begin()
    yfc = dense.linear(xf, fc.weight, fc.bias, relu=True)
    yfc.square().sum().backward()
    prof = capi.profile_end()
    assert 'htd_conv2d_fwd_x3h' in prof and 'htd_conv2d_bwd_weight_h2' in prof and 'htd_conv2d_bwd_data_x3h' in prof, sorted(prof)
    dense.new_step()


def test_fused_roi_tiles_leave_their_maximum(h2):
    """htd_fuse_global_fwd_amax / htd_plain_and_fused_fwd_amax: the outputs of the plain entry points and their largest magnitude bit
    for bit (what the heads' first FC layers scale by); through mmcv_ops the outputs come back tagged."""
    from htd_amd import capi, dense, mmcv_ops
    dev = torch.device('cuda:0')
    torch.manual_seed(11)
    n, C, P, B = 333, 64, 49, 3
    x = (torch.randn(n, C, 7, 7, device=dev) * 4).contiguous(memory_format=CL)
    rois = torch.cat([torch.randint(0, B, (n, 1), device=dev).float(), torch.rand(n, 4, device=dev) * 100], 1)
    g = torch.randn(B, C, device=dev) * 9
    extra = torch.randn_like(x).contiguous(memory_format=CL)
    S = capi.current_stream_ptr
    for e in (None, extra):
        a, b, slot = torch.empty_like(x), torch.empty_like(x), torch.zeros(1, device=dev)
        capi.call('htd_fuse_global_fwd', capi.ptr(x), capi.ptr(rois), capi.ptr(g), capi.ptr(e), 0.5, capi.ptr(a), n, P, C, B, S())
        capi.call('htd_fuse_global_fwd_amax', capi.ptr(x), capi.ptr(rois), capi.ptr(g), capi.ptr(e), 0.5, capi.ptr(b), n, P, C, B,
                  capi.ptr(slot), S())
        assert torch.equal(a, b) and float(slot) == float(a.abs().max())
    a = torch.empty(2 * n, C, 7, 7, device=dev).contiguous(memory_format=CL)
    b, slot = torch.empty_like(a), torch.zeros(1, device=dev)
    capi.call('htd_plain_and_fused_fwd', capi.ptr(x), capi.ptr(rois), capi.ptr(g), capi.ptr(a), n, P, C, B, S())
    capi.call('htd_plain_and_fused_fwd_amax', capi.ptr(x), capi.ptr(rois), capi.ptr(g), capi.ptr(b), n, P, C, B, capi.ptr(slot), S())
    assert torch.equal(a, b) and float(slot) == float(a.abs().max())
    x[5, 3, 2, 1] = float('nan')
    slot.zero_()
    capi.call('htd_plain_and_fused_fwd_amax', capi.ptr(x), capi.ptr(rois), capi.ptr(g), capi.ptr(b), n, P, C, B, capi.ptr(slot), S())
    assert torch.isnan(slot).item()
    x[5, 3, 2, 1] = 0.0
    dense.new_step()
    out = mmcv_ops.fuse_global(x, rois, g.view(B, C, 1, 1))
    both = mmcv_ops.PlainAndFusedFunction.apply(x, rois, g.view(B, C, 1, 1))
    for t in (out, both):
        am = dense.carried_amax(t.permute(0, 2, 3, 1).reshape(t.size(0), -1))
        assert am is not None and float(am) == float(t.abs().max())
    dense.new_step()


@pytest.mark.parametrize('B,H,W,Ci,Co,k,stride,relu,res', [
    (2, 40, 56, 128, 128, 3, 2, 1, False),        # the stride-2 3x3 of a stage's first block
    (4, 100, 168, 64, 64, 3, 2, 0, False),
    (4, 50, 84, 256, 1024, 1, 1, 0, False),       # 1056 tiles of 128x128: the balanced tail and its epilogue kernel
    (1, 13, 21, 256, 256, 3, 1, 1, True),         # few tiles: split-K and its epilogue kernel
    (2, 25, 33, 64, 78, 1, 1, 0, False),          # Co % 4 != 0: the scalar epilogue
    (2, 31, 45, 64, 128, 1, 2, 0, False),         # stride-2 1x1 shortcut
])
def test_igemm_kernels_leave_their_maximum(h2, B, H, W, Ci, Co, k, stride, relu, res):
    """htd_conv2d_fwd_amax / htd_conv2d_bwd_data_amax: outputs of the plain entry points, and their largest magnitude bit for bit
    (every epilogue of conv_igemm_kernel: in-kernel, split-K, balanced tail, scalar columns; the strided data gradient's parity
    classes raise one scalar)."""
    from htd_amd import capi, dense
    dev = torch.device('cuda:0')
    torch.manual_seed(B * H + Co)
    pad = k // 2
    x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.1).contiguous(memory_format=CL)
    bias = torch.randn(Co, device=dev)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    r = torch.randn(B, Co, Ho, Wo, device=dev).contiguous(memory_format=CL) if res else None
    S = capi.current_stream_ptr
    ws = dense._splitk_ws(B * Ho * Wo, Co, Ci, k, k, dev)
    y0, y1, slot = [torch.empty(B, Co, Ho, Wo, device=dev).contiguous(memory_format=CL) for _ in range(2)] + [torch.zeros(1, device=dev)]
    capi.call('htd_conv2d_fwd', capi.ptr(x), capi.ptr(w), capi.ptr(bias), capi.ptr(r), 0, 0, capi.ptr(y0), B, H, W, Ci, Co, k, k, stride,
              pad, 1, relu, capi.ptr(ws), S())
    capi.call('htd_conv2d_fwd_amax', capi.ptr(x), capi.ptr(w), capi.ptr(bias), capi.ptr(r), 0, 0, capi.ptr(y1), capi.ptr(slot), B, H, W,
              Ci, Co, k, k, stride, pad, 1, relu, capi.ptr(ws), S())
    assert torch.equal(y0, y1) and float(slot) == float(y0.abs().max())
    if Co % 8:
        return
    # data gradient of the same layer: gy (B, Co, Ho, Wo) -> gx (B, Ci, H, W), masked by x > 0
    gy = torch.randn(B, Co, Ho, Wo, device=dev).contiguous(memory_format=CL)
    wT = torch.empty(Ci * k * k * Co, device=dev)
    capi.call('htd_conv2d_flip_weights', capi.ptr(w), capi.ptr(wT), Co, k, k, Ci, S())
    ws = dense._splitk_ws(B * H * W, Ci, Co, k, k, dev)
    g0, g1 = [torch.empty(B, Ci, H, W, device=dev).contiguous(memory_format=CL) for _ in range(2)]
    slot.zero_()
    capi.call('htd_conv2d_bwd_data', capi.ptr(gy), capi.ptr(wT), capi.ptr(x), None, capi.ptr(g0), B, H, W, Ci, Co, k, k, stride, pad, 1,
              capi.ptr(ws), S())
    capi.call('htd_conv2d_bwd_data_amax', capi.ptr(gy), capi.ptr(wT), capi.ptr(x), None, capi.ptr(g1), capi.ptr(slot), B, H, W, Ci, Co, k,
              k, stride, pad, 1, capi.ptr(ws), S())
    assert torch.equal(g0, g1) and float(slot) == float(g0.abs().max())
    # through dense: the outputs come back tagged
    dense.new_step()
    y = dense._fwd_raw(x, w, bias, r, stride, pad, 1, bool(relu))
    gx = dense._dgrad_raw(gy, w, x.shape, stride, pad, 1, mask_src=x)
    for t in (y, gx):          # (whichever kernel dense gives the layer to)
        am = dense.carried_amax(t)
        assert am is not None and float(am) == float(t.abs().max())
    dense.new_step()


def test_max_pool_leaves_its_maximum(h2):
    from htd_amd import capi, dense, mmcv_ops
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    x = torch.randn(2, 64, 57, 83, device=dev).contiguous(memory_format=CL)
    dense.new_step()
    y = mmcv_ops.MaxPool2dFunction.apply(x, 3, 2, 1)
    assert torch.equal(y, F.max_pool2d(x, 3, 2, 1))
    am = dense.carried_amax(y)
    assert am is not None and float(am) == float(y.abs().max())
    x[1, 7, 20, 30] = float('nan')
    y = mmcv_ops.MaxPool2dFunction.apply(x, 3, 2, 1)
    assert torch.isnan(dense.carried_amax(y)).item()
    dense.new_step()


@pytest.mark.parametrize('B,H,W,Ci,Co', [(2, 40, 56, 128, 128), (1, 33, 47, 64, 96), (3, 25, 42, 256, 512), (2, 7, 9, 48, 64)])
def test_strided_3x3_on_the_tap_list_loop(h2, B, H, W, Ci, Co):
    """The stride-2 3x3 layer of a stage's first block on conv_x3p_kernel's 1x1 loop (tap-list mode, H2): integer-exact on small
    integers -- odd sizes, borders, bias, residual, ReLU -- the maximum it leaves is its output's, and on wide-range data its error
    against fp64 is within RMS_BOUND of the fp32-input matrix instructions' (conv_igemm_kernel with set_math(0))."""
    from htd_amd import capi, dense
    dev = torch.device('cuda:0')
    if not h2.htd_conv2d_x3h_strided_supported(Ci, Co, 3, 3, 2, 1, 1):
        pytest.skip('strided H2 path switched off')
    torch.manual_seed(H * W + Ci)
    xi = torch.randint(-4, 5, (B, Ci, H, W), device=dev).float().contiguous(memory_format=CL)
    wi = torch.randint(-2, 3, (Co, Ci, 3, 3), device=dev).float().contiguous(memory_format=CL)
    bi = torch.randint(-8, 9, (Co, ), device=dev).float()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ri = torch.randint(-16, 17, (B, Co, Ho, Wo), device=dev).float().contiguous(memory_format=CL)
    dense.new_step()
    dense.tag_amax(xi, dense.absmax(xi))
    capi.profile_begin()
    y = dense._fwd_raw(xi, wi, bi, ri, 2, 1, 1, True)
    prof = capi.profile_end()
    assert 'htd_conv2d_fwd_x3h' in prof and 'htd_conv2d_fwd' not in prof, sorted(prof)
    ref = torch.relu(F.conv2d(xi.double(), wi.double(), bi.double(), 2, 1) + ri.double())
    assert torch.equal(y.double(), ref)
    am = dense.carried_amax(y)
    assert am is not None and float(am) == float(y.abs().max())
    # wide dynamic range
    x = (torch.randn(B, Ci, H, W, device=dev) * torch.exp2(torch.randint(-6, 7, (B, Ci, 1, 1), device=dev).float())).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05 * torch.exp2(torch.randint(-3, 4, (Co, 1, 1, 1), device=dev).float())).contiguous(memory_format=CL)
    ref = F.conv2d(x.double(), w.double(), None, 2, 1)
    dense.new_step()
    dense.tag_amax(x, dense.absmax(x))
    y = dense._fwd_raw(x, w, None, None, 2, 1, 1, False)
    h2.htd_conv2d_set_h2(0)
    h2.htd_conv2d_set_math(0)
    yn = dense._fwd_raw(x, w, None, None, 2, 1, 1, False)
    h2.htd_conv2d_set_math(1)
    h2.htd_conv2d_set_h2(1)
    e, en = (y.double() - ref), (yn.double() - ref)
    assert float(e.square().mean().sqrt()) <= RMS_BOUND * float(en.square().mean().sqrt()) + 1e-12
    assert float(e.abs().max()) <= MAX_BOUND * 2 * float(en.abs().max()) + 1e-12
    dense.new_step()


@pytest.mark.parametrize('B,H,W,Ci,Co,k', [(2, 40, 56, 128, 128, 3), (1, 33, 47, 64, 96, 3), (3, 25, 42, 256, 512, 3), (2, 7, 9, 48, 64, 3),
                                            (2, 31, 45, 64, 128, 1), (4, 50, 84, 512, 1024, 1), (1, 1, 1, 64, 64, 3)])
def test_strided_data_gradient_on_the_tap_list_loop(h2, B, H, W, Ci, Co, k):
    """htd_conv2d_bwd_data_x3h_strided: the stride-2 layers' data gradient as one H2 launch per parity class of gx's pixels (tap lists,
    strided output map): integer-exact on small integers with the producer's ReLU mask, odd sizes and borders; the maximum it
    leaves is gx's; error against fp64 within RMS_BOUND of the fp32-input matrix instructions' on wide-range data."""
    from htd_amd import capi, dense
    dev = torch.device('cuda:0')
    pad = k // 2
    if not h2.htd_conv2d_bwd_data_x3h_strided_supported(Ci, Co, k, k, 2, pad, 1):
        pytest.skip('strided H2 path switched off')
    torch.manual_seed(H * W + Co + k)
    Ho, Wo = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    gi = torch.randint(-4, 5, (B, Co, Ho, Wo), device=dev).float().contiguous(memory_format=CL)
    wi = torch.randint(-2, 3, (Co, Ci, k, k), device=dev).float().contiguous(memory_format=CL)
    mask = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=CL)
    dense.new_step()
    dense.tag_amax(gi, dense.absmax(gi))
    capi.profile_begin()
    gx = dense._dgrad_raw(gi, wi, (B, Ci, H, W), 2, pad, 1, mask_src=mask)
    prof = capi.profile_end()
    assert 'htd_conv2d_bwd_data_x3h' in prof and 'htd_conv2d_bwd_data' not in prof, sorted(prof)
    ref = torch.nn.grad.conv2d_input((B, Ci, H, W), wi.double(), gi.double(), 2, pad) * (mask > 0)
    assert torch.equal(gx.double(), ref)
    am = dense.carried_amax(gx)
    assert am is not None and float(am) == float(gx.abs().max())
    g = (torch.randn(B, Co, Ho, Wo, device=dev) * torch.exp2(torch.randint(-6, 7, (B, Co, 1, 1), device=dev).float())).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05 * torch.exp2(torch.randint(-3, 4, (1, Ci, 1, 1), device=dev).float())).contiguous(memory_format=CL)
    ref = torch.nn.grad.conv2d_input((B, Ci, H, W), w.double(), g.double(), 2, pad)
    dense.new_step()
    dense.tag_amax(g, dense.absmax(g))
    gx = dense._dgrad_raw(g, w, (B, Ci, H, W), 2, pad, 1)
    h2.htd_conv2d_set_h2(0)
    h2.htd_conv2d_set_math(0)
    gn = dense._dgrad_raw(g, w, (B, Ci, H, W), 2, pad, 1)
    h2.htd_conv2d_set_math(1)
    h2.htd_conv2d_set_h2(1)
    e, en = (gx.double() - ref), (gn.double() - ref)
    assert float(e.square().mean().sqrt()) <= RMS_BOUND * float(en.square().mean().sqrt()) + 1e-12
    assert float(e.abs().max()) <= MAX_BOUND * 2 * float(en.abs().max()) + 1e-12
    dense.new_step()
