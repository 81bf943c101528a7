#!/usr/bin/env python3
"""Static-shape train path against the per-image path on the 'no_gt_image_and_few_proposals' scenario of
tests/test_gpu_detector.py: the losses of both, the parameters whose gradients differ most, and for the FC layer where a ReLU flipped
which output units differ.  With DBG_HOOK=1 every H2 launch of both passes is re-run on the six-product form and compared (weight
gradients) or has the maximum its epilogue left checked against its output (forward, data gradients).
usage: [DBG_HOOK=0] [HTD_H2_EMIT_OFF=...] python tools/static_path_diff.py"""
import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import numpy as np, torch, ctypes
import test_gpu_detector as TD
from golden_util import load_seeded_
from htd_amd import capi, dense
from htd_amd.configs import build_htd_detector
from htd_amd.core import set_randperm
from htd_amd.core.bbox import set_sample_keys
g = np.load(ROOT+'/tests/golden/detector.npz', allow_pickle=True)
dev = torch.device('cuda:0')
det = build_htd_detector(cfg=TD.small_cfg()); load_seeded_(det, 'det.'); det = det.to(dev).train()
img, metas, gts, labels = TD.inputs(g, dev)
gts, labels = [gts[0], gts[1][:0]], [labels[0], labels[1][:0]]
det.train_cfg.rpn_proposal.nms_post = 30
coef = torch.tensor([12.9898, 78.233, 37.719, 93.989], device=dev)
set_sample_keys(lambda cand: torch.frac(torch.sin((cand * coef).sum(-1)) * 43758.5453).abs())
set_randperm(None)
orig = capi.call
def ptrval(p): return None if p is None else p.value
hip = ctypes.CDLL('libamdhip64.so')
def dev_copy(ptr, n):
    buf = torch.empty(n, device=dev)
    hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ptr, ctypes.c_size_t(n*4), 3)
    return buf
def call(name, *args, **kw):
    before = None
    if name == 'htd_conv2d_bwd_weight_h2':
        x, gy, ax, ag, gw, gb, B,H,W,Ci,Co,kh,kw_,st,pad,dil, acc, ws, s = args
        if acc: before = dev_copy(gw, Co*kh*kw_*Ci)
    orig(name, *args, **kw)
    if name == 'htd_conv2d_bwd_weight_h2':
        n = Co*kh*kw_*Ci
        ref = torch.empty(n, device=dev); refb = torch.empty(Co, device=dev)
        orig('htd_conv2d_bwd_weight', x, gy, capi.ptr(ref), capi.ptr(refb) if gb is not None else None, B,H,W,Ci,Co,kh,kw_,st,pad,dil, ws, s)
        buf = dev_copy(gw, n)
        if acc: buf = buf - before
        err = float((buf-ref).abs().max() / ref.abs().max().clamp_min(1e-20))
        print(f'  wgrad_h2 B={B} H={H} W={W} Ci={Ci} Co={Co} k={kh} acc={acc} relerr {err:.2e} amax_x {float(dev_copy(ax,1)):.3e} amax_g {float(dev_copy(ag,1)):.3e} refmax {float(ref.abs().max()):.3e}')
    if name == 'htd_conv2d_bwd_data_x3h':
        gy, am, wp, mask, accum, gx, gxp, oslot, B,H,W,Ci,Co,kh,kw_,pad, ws, s = args
        print(f'  dgrad_h2 B={B} H={H} W={W} Ci={Ci} Co={Co} k={kh} amax {float(dev_copy(am,1)):.3e} outmax {float(dev_copy(gx, B*H*W*Ci).abs().max()):.3e} slot {float(dev_copy(oslot,1)):.3e}')
    if name == 'htd_conv2d_fwd_x3h':
        x, am, wp, bias, res, rh, rw, y, yp, oslot, B,H,W,Ci,Co,kh,kw_,st,pad,relu, ws, s = args
        Ho, Wo = (H + 2*pad - kh)//st + 1, (W + 2*pad - kw_)//st + 1
        om = float(dev_copy(y, B*Ho*Wo*Co).abs().max()); sl = float(dev_copy(oslot,1))
        if om != sl: print(f'  FWD slot mismatch B={B} H={H} W={W} Ci={Ci} Co={Co} k={kh}: out max {om:.6e} slot {sl:.6e}')
if os.environ.get('DBG_HOOK', '1') == '1':
    capi.call = call
    dense.capi.call = call
out = {}
for static in (True, False):
    print('=== static', static)
    det.roi_head.static_shapes = static
    det.zero_grad()
    losses = det(img=img, img_metas=metas, gt_bboxes=gts, gt_labels=labels)
    loss, lv = det._parse_losses(losses)
    print('LOSSES', {k: float(v) for k, v in lv.items()})
    loss.backward()
    torch.cuda.synchronize()
    out[static] = {n: p.grad.detach().clone() for n, p in det.named_parameters() if p.grad is not None}
errs = []
for n in out[True]:
    if n in out[False]:
        sc = float(out[False][n].abs().max())
        if sc > 0: errs.append((float((out[True][n] - out[False][n]).abs().max()) / sc, n))
errs.sort(reverse=True)
print('TOP', errs[:6])
for n in ('roi_head.bbox_head.1.fcs.2.weight', 'roi_head.bbox_head.1.fcs.2.bias'):
    a, b = out[True][n], out[False][n]
    d = (a - b).abs() / b.abs().max()
    bad = (d > 2e-4).nonzero()
    print(n, tuple(a.shape), 'elements over 2e-4:', bad.shape[0], 'rows', sorted(set(bad[:, 0].tolist()))[:10])

