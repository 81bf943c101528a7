"""Activation planes (round 4; csrc/conv_x3.hip: X3Params::xp / yp, conv_x3q_kernel, act_planes_kernel).

A 1x1 / stride-1 layer reads its input as three pre-split bf16 planes that the PRODUCER of the map wrote from its epilogue.
The split is exact arithmetic and conv_x3q_kernel issues conv_x3p_kernel's products in conv_x3p_kernel's order, so everything
here is bit-for-bit: planes against the tensor formulation of the split, emitted planes against htd_act_planes of the stored
map, plane-fed convolutions against the fp32-fed ones.  (Role in the reference: cuDNN behind backbones/resnet.py:260-300.)"""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _planes_ref(x):
    """(B,C,H,W) fp32 -> int32 words of the plane image [C/16][6][rows][8 bf16]: a0 = bf16(a), a1 = bf16(a - a0),
    a2 = bf16(a - a0 - a1), round to nearest even, both differences exact in fp32 (conv_fwd.hip split3x2)."""
    from htd_amd import capi
    B, C, H, W = x.shape
    M = B * H * W
    rows = capi.lib().htd_act_planes_rows(M)
    a = x.permute(0, 2, 3, 1).reshape(M, C).float()
    a0 = a.bfloat16()
    r1 = a - a0.float()
    a1 = r1.bfloat16()
    a2 = (r1 - a1.float()).bfloat16()
    out = torch.zeros(C // 16, 6, rows, 8, dtype=torch.bfloat16, device=x.device)
    for q, pl in enumerate((a0, a1, a2)):
        v = pl.reshape(M, C // 16, 2, 8).permute(1, 2, 0, 3)          # [cs][half][m][8]
        out[:, 2 * q:2 * q + 2, :M] = v
    return out, M, rows


def _valid_words(planes, C, M, rows):
    """the rows < M of a plane buffer (int32 words) as [C/16][6][M][4]"""
    return planes.view(C // 16, 6, rows, 4)[:, :, :M]


@pytest.mark.parametrize('B,C,H,W', [(1, 16, 1, 1), (2, 64, 9, 13), (3, 256, 7, 7), (1, 48, 130, 3)])
def test_act_planes_kernel_is_the_exact_three_way_split(B, C, H, W):
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B * 7 + C)
    x = (torch.randn(B, C, H, W, generator=g) * torch.exp(4 * torch.randn(B, C, H, W, generator=g))).to(dev).contiguous(memory_format=CL)
    ref, M, rows = _planes_ref(x)
    got = dense.act_planes(x)
    assert torch.equal(_valid_words(got, C, M, rows), ref.view(torch.int32).view(C // 16, 6, rows, 4)[:, :, :M])
    # the three planes add up to the fp32 value exactly (8 + 8 + 8 significand bits)
    pl = got.view(torch.bfloat16).view(C // 16, 3, 2, rows, 8)[:, :, :, :M].float().sum(1)        # [cs][half][m][8]
    back = pl.permute(2, 0, 1, 3).reshape(M, C)
    assert torch.equal(back, x.permute(0, 2, 3, 1).reshape(M, C))


PLANE_CASES = [
    # B, Cmid, H, W, Cout: a 3x3 layer (Cmid -> Cmid, bias + ReLU) feeding a 1x1 layer (Cmid -> Cout, bias + residual + ReLU);
    # ragged tiles in M and in both channel counts, K-range plans (few tiles, long K), one-pixel maps
    (2, 64, 20, 28, 256), (1, 256, 13, 17, 1024), (4, 128, 25, 21, 512), (1, 16, 1, 1, 64), (3, 48, 9, 11, 160),
    (1, 512, 25, 42, 2048), (2, 32, 40, 40, 96),
]


@pytest.mark.parametrize('B,Cm,H,W,Co', PLANE_CASES)
def test_emitted_planes_and_plane_fed_1x1_layers_are_bit_identical(B, Cm, H, W, Co):
    """conv2 -> conv3 of a bottleneck and the mirrored pair of its backward: the planes the 3x3 epilogue (or its K-range reduce
    pass) writes equal htd_act_planes of the map it stores; the 1x1 layer fed with them returns the bits of the fp32-fed call,
    forward (bias, residual, ReLU) and data gradient (mask_src, accum), and its own emitted planes are right too."""
    from htd_amd import capi, dense
    L = capi.lib()
    if not (L.htd_conv2d_x3p_supported(Cm, Cm, 3, 3, 1, 1, 1) and L.htd_conv2d_x3p_supported(Cm, Co, 1, 1, 1, 0, 1)):
        pytest.skip('conv_x3p_kernel switched off')
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B + Cm + Co + H)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    x = rn(B, Cm, H, W).contiguous(memory_format=CL)
    w2 = (rn(Cm, Cm, 3, 3) / (Cm * 9) ** 0.5).contiguous(memory_format=CL)
    w3 = (rn(Co, Cm, 1, 1) / Cm ** 0.5).contiguous(memory_format=CL)
    b2, b3 = rn(Cm), rn(Co)
    res = rn(B, Co, H, W).contiguous(memory_format=CL)
    M = B * H * W
    rows = L.htd_act_planes_rows(M)
    dense.new_step()
    # forward: 3x3 emits, 1x1 consumes (and emits its own)
    h_plain = dense._fwd_raw(x, w2, b2, None, 1, 1, 1, True)
    h, hp = dense._fwd_raw(x, w2, b2, None, 1, 1, 1, True, emit=True)
    assert hp is not None and torch.equal(h, h_plain)
    assert torch.equal(_valid_words(hp, Cm, M, rows), _valid_words(dense.act_planes(h), Cm, M, rows))
    y_plain = dense._fwd_raw(h, w3, b3, res, 1, 0, 1, True)
    y, yp = dense._fwd_raw(h, w3, b3, res, 1, 0, 1, True, x_planes=hp, emit=True)
    assert torch.equal(y, y_plain)
    if Co % 16 == 0:
        assert torch.equal(_valid_words(yp, Co, M, rows), _valid_words(dense.act_planes(y), Co, M, rows))
    # backward: the 3x3 data gradient emits, the 1x1 data gradient (Cm -> Co channels) consumes
    if L.htd_conv2d_x3p_supported(Cm, Co, 1, 1, 1, 0, 1):
        w1 = (rn(Cm, Co, 1, 1) / Co ** 0.5).contiguous(memory_format=CL)          # conv1 of a block: Co -> Cm
        gy = rn(B, Cm, H, W).contiguous(memory_format=CL)
        gm_plain = dense._dgrad_raw(gy, w2, (B, Cm, H, W), 1, 1, 1, mask_src=x)
        gm, gmp = dense._dgrad_raw(gy, w2, (B, Cm, H, W), 1, 1, 1, mask_src=x, emit=True)
        assert gmp is not None and torch.equal(gm, gm_plain)
        assert torch.equal(_valid_words(gmp, Cm, M, rows), _valid_words(dense.act_planes(gm), Cm, M, rows))
        acc, msk = rn(B, Co, H, W).contiguous(memory_format=CL), rn(B, Co, H, W).contiguous(memory_format=CL)
        gx_plain = dense._dgrad_raw(gm, w1, (B, Co, H, W), 1, 0, 1, mask_src=msk, accum=acc)
        gx = dense._dgrad_raw(gm, w1, (B, Co, H, W), 1, 0, 1, mask_src=msk, accum=acc, g_planes=gmp)
        assert torch.equal(gx, gx_plain)
    dense.new_step()


def test_plane_fed_layer_is_exact_on_integers():
    """small-integer operands: every product and partial sum is exact, so the plane-fed 1x1 layer equals the fp64 result"""
    from htd_amd import dense
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    for B, Ci, H, W, Co in [(2, 256, 20, 28, 512), (1, 1024, 9, 7, 96), (5, 64, 3, 3, 1024)]:
        xi = torch.randint(-4, 5, (B, Ci, H, W), generator=g).float()
        wi = torch.randint(-3, 4, (Co, Ci, 1, 1), generator=g).float()
        ref = F.conv2d(xi.double(), wi.double())
        x = xi.to(dev).contiguous(memory_format=CL)
        w = wi.to(dev).contiguous(memory_format=CL)
        dense.new_step()
        y = dense._fwd_raw(x, w, None, None, 1, 0, 1, False, x_planes=dense.act_planes(x))
        assert torch.equal(y.cpu().double(), ref)
    dense.new_step()


_VARIANTS = r'''
import os, sys, torch
sys.path.insert(0, %r)
from htd_amd import capi, dense
CL = torch.channels_last
L = capi.lib()
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
bad = []
for B, Ci, H, W, Co in [(2, 128, 20, 28, 512), (1, 1024, 13, 17, 256), (4, 256, 25, 21, 1024), (3, 48, 9, 11, 160)]:
    x = torch.randn(B, Ci, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    w = (torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5).to(dev).contiguous(memory_format=CL)
    b = torch.randn(Co, generator=g).to(dev)
    r = torch.randn(B, Co, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    xp = dense.act_planes(x)
    os.environ.pop('HTD_X3P_FORCE_TILE', None)
    for tile in (0, 1, 2, 3):
        os.environ['HTD_X3P_FORCE_TILE'] = str(tile)
        os.environ['HTD_X3P_MFMA'] = '32'
        ref32 = dense._fwd_raw(x, w, b, r, 1, 0, 1, True)
        os.environ['HTD_X3P_MFMA'] = '16'
        ref16 = dense._fwd_raw(x, w, b, r, 1, 0, 1, True)
        for mf, ref in ((32, ref32), (16, ref16)):
            for ns in (2, 3):
                os.environ['HTD_X3Q_MFMA'], os.environ['HTD_X3Q_NS'] = str(mf), str(ns)
                y = dense._fwd_raw(x, w, b, r, 1, 0, 1, True, x_planes=xp)
                if not torch.equal(y, ref):
                    bad.append((B, Ci, H, W, Co, tile, mf, ns, float((y - ref).abs().max())))
print('BAD', bad)
sys.exit(1 if bad else 0)
'''


def test_every_instantiation_of_the_plane_fed_kernel_matches_its_twin():
    """All 16 conv_x3q_kernel instantiations (four tiles x two MFMA shapes x two / three LDS stages) against the conv_x3p_kernel
    instantiation with the same tile and MFMA shape: bit-identical outputs.  Tune mode (re-read environment per call) is a
    load-time switch of the library, hence the child process."""
    env = dict(os.environ, HTD_X3P_TUNE='1')
    r = subprocess.run([sys.executable, '-c', _VARIANTS % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_res_stage_with_and_without_planes_is_bit_identical():
    """dense.ResStageFunction threads the planes conv2 -> conv3 (forward) and conv2's data gradient -> conv1's (backward).  With
    the thresholds lowered so that every block of a small stage uses them, outputs and ALL gradients must equal the run without
    planes bit for bit."""
    import torch.nn as nn
    from htd_amd import dense
    from htd_amd.detector.resnet import Bottleneck, ResLayer
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    layer = ResLayer(Bottleneck, 256, 64, 3, stride=1, norm_cfg=dict(type='BN', requires_grad=True)).to(dev)
    for m in layer.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    layer.eval()                                  # frozen-BN statistics, parameters still trainable (norm_eval)
    x0 = torch.randn(2, 256, 24, 40, device=dev).contiguous(memory_format=CL)
    g = torch.randn(2, 256, 24, 40, device=dev).contiguous(memory_format=CL)
    saved = (dense.ACT_PLANES, dense.ACT_PLANES_MIN_TILES, dense.ACT_PLANES_MIN_ROWS)
    out = {}
    h2_was = dense.capi.lib().htd_conv2d_set_h2(0)         # the planes belong to the three-piece bf16 form (H2 layers carry a maximum instead)
    try:
        for on in (False, True):
            dense.ACT_PLANES, dense.ACT_PLANES_MIN_TILES, dense.ACT_PLANES_MIN_ROWS = on, 1, 0
            dense.new_step()
            layer.zero_grad()
            x = x0.clone().requires_grad_()
            calls = []
            orig = dense.capi.call

            def spy(name, *a, **k):
                if name in ('htd_conv2d_fwd_x3q', 'htd_conv2d_bwd_data_x3q') and a[1] is not None:
                    calls.append(name)                    # the plane-fed form: the input planes operand is there
                return orig(name, *a, **k)
            dense.capi.call = spy
            try:
                y = layer(x)
                y.backward(g)
            finally:
                dense.capi.call = orig
            assert ('htd_conv2d_fwd_x3q' in calls) == on and ('htd_conv2d_bwd_data_x3q' in calls) == on
            out[on] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters()})
    finally:
        dense.ACT_PLANES, dense.ACT_PLANES_MIN_TILES, dense.ACT_PLANES_MIN_ROWS = saved
        dense.capi.lib().htd_conv2d_set_h2(h2_was)
        dense.new_step()
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
    for n in out[True][2]:
        assert torch.equal(out[True][2][n], out[False][2][n]), n
