"""oracle/ops.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes bindings of oracle/c/htd_oracle_ops.c (plain-C CPU restatement of the
mmcv-full 1.2.1 operators the HTD path calls: RoIAlign, nms, soft_nms,
deformable conv) plus the thin Python wrappers mmcv puts around them
(``batched_nms``'s coordinate-offset trick).  PARITY UNPINNED for these ops:
the reference holds no numeric vectors for them.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  CPU tensors only.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libhtd_oracle.so')
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, 'c', 'htd_oracle_ops.c')
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(_HERE, 'c'), '-s'])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_nms.restype = ctypes.c_int64
        _lib.oracle_soft_nms.restype = ctypes.c_int64
    return _lib


def _fp(t):
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


def _ip(t):
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_int64))


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


# --------------------------------------------------------------------- RoIAlign
def roi_align_fwd(feat, rois, out_size=7, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    feat, rois = _f32c(feat), _f32c(rois)
    B, C, H, W = feat.shape
    n = rois.shape[0]
    ph, pw = (out_size, out_size) if isinstance(out_size, int) else out_size
    out = torch.zeros(n, C, ph, pw)
    lib().oracle_roi_align_fwd(_fp(feat), _fp(rois), _fp(out), n, C, H, W, ph, pw,
                               ctypes.c_float(spatial_scale), int(sampling_ratio), int(aligned))
    return out


def roi_align_bwd(grad_out, rois, feat_shape, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    grad_out, rois = _f32c(grad_out), _f32c(rois)
    B, C, H, W = feat_shape
    n, _, ph, pw = grad_out.shape
    gin = torch.zeros(B, C, H, W)
    lib().oracle_roi_align_bwd(_fp(grad_out), _fp(rois), _fp(gin), n, C, H, W, ph, pw,
                               ctypes.c_float(spatial_scale), int(sampling_ratio), int(aligned))
    return gin


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, out_size, spatial_scale, sampling_ratio, aligned):
        ctx.save_for_backward(rois)
        ctx.args = (tuple(feat.shape), spatial_scale, sampling_ratio, aligned)
        return roi_align_fwd(feat, rois, out_size, spatial_scale, sampling_ratio, aligned)

    @staticmethod
    def backward(ctx, g):
        rois, = ctx.saved_tensors
        shape, scale, sr, al = ctx.args
        return roi_align_bwd(g, rois, shape, scale, sr, al), None, None, None, None, None


def roi_align(feat, rois, out_size=7, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    """Differentiable (w.r.t. feat) CPU RoIAlign; mmcv.ops.roi_align argument meaning."""
    return _RoIAlignFn.apply(feat, rois, out_size, spatial_scale, sampling_ratio, aligned)


class RoIAlign(torch.nn.Module):
    """Signature of mmcv.ops.RoIAlign as the reference constructs it
    (roi_extractors/base_roi_extractor.py:49-56): aligned=True, avg pooling."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg',
                 aligned=True, use_torchvision=False):
        super().__init__()
        assert pool_mode == 'avg'
        self.output_size = (output_size, output_size) if isinstance(output_size, int) \
            else tuple(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.aligned = aligned

    def forward(self, feat, rois):
        return roi_align(feat, rois, self.output_size, self.spatial_scale, self.sampling_ratio,
                         self.aligned)


# --------------------------------------------------------------------- NMS
def nms(boxes, scores, iou_threshold, offset=0):
    """-> (dets (k,5), keep (k,) int64 in descending score order)."""
    boxes, scores = _f32c(boxes), _f32c(scores)
    n = boxes.shape[0]
    keep = torch.zeros(n, dtype=torch.int64)
    k = lib().oracle_nms(_fp(boxes), _fp(scores), ctypes.c_int64(n), ctypes.c_float(iou_threshold),
                         int(offset), _ip(keep)) if n else 0
    keep = keep[:k]
    dets = torch.cat([boxes[keep], scores[keep].reshape(-1, 1)], dim=1)
    return dets, keep


def soft_nms(boxes, scores, iou_threshold=0.3, sigma=0.5, min_score=1e-3, method='linear', offset=0,
             iou_thr=None):
    if iou_thr is not None:  # deprecated spelling still used by configs/htd/htd_resnet101_2x.py:298
        iou_threshold = iou_thr
    code = {'naive': 0, 'linear': 1, 'gaussian': 2}[method]
    boxes, scores = _f32c(boxes), _f32c(scores)
    n = boxes.shape[0]
    dets = torch.zeros(n, 5)
    inds = torch.zeros(n, dtype=torch.int64)
    k = lib().oracle_soft_nms(_fp(boxes), _fp(scores), ctypes.c_int64(n), ctypes.c_float(iou_threshold),
                              ctypes.c_float(sigma), ctypes.c_float(min_score), code, int(offset),
                              _fp(dets), _ip(inds)) if n else 0
    return dets[:k], inds[:k]


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms (1.2.1): per-class NMS through a coordinate offset;
    above split_thr boxes the classes are processed one by one."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + 1)
        boxes_for_nms = boxes + offsets[:, None]
    op = {'nms': nms, 'soft_nms': soft_nms}[cfg.pop('type', 'nms')]
    split_thr = cfg.pop('split_thr', 10000)
    if len(boxes_for_nms) < split_thr:
        dets, keep = op(boxes_for_nms, scores, **cfg)
        boxes = boxes[keep]
        scores = dets[:, -1]
    else:
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        for cid in torch.unique(idxs):
            mask = (idxs == cid).nonzero(as_tuple=False).view(-1)
            dets, keep = op(boxes_for_nms[mask], scores[mask], **cfg)
            total_mask[mask[keep]] = True
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        keep = keep[scores[keep].argsort(descending=True, stable=True)]
        boxes = boxes[keep]
        scores = scores[keep]
    return torch.cat([boxes, scores[:, None]], -1), keep


# --------------------------------------------------------------------- DCN
def deform_conv2d(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deform_groups=1,
                  mask=None):
    """Forward only (the reference has no CPU DCN at all:
    build/lib/mmdet/ops/dcn/deform_conv.py:45-46)."""
    x, offset, weight = _f32c(x), _f32c(offset), _f32c(weight)
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    p = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    (sh, sw), (ph, pw), (dh, dw) = p(stride), p(padding), p(dilation)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    assert offset.shape == (B, 2 * deform_groups * kh * kw, Ho, Wo), offset.shape
    out = torch.zeros(B, Co, Ho, Wo)
    mptr = None
    if mask is not None:
        mask = _f32c(mask)
        mptr = _fp(mask)
    lib().oracle_deform_conv_fwd(_fp(x), _fp(offset), mptr, _fp(weight), _fp(out), B, C, H, W, Co, kh, kw,
                                 sh, sw, ph, pw, dh, dw, groups, deform_groups)
    return out


def deform_conv2d_autograd(x, offset, weight, stride=1, padding=0, dilation=1, mask=None, groups=1):
    """Differentiable pure-torch restatement of the same DCN (deform_groups=1; `groups` conv groups as in the
    ResNeXt bottleneck), used to check gradients of the HIP kernels.  Bilinear sampling with
    zero padding, written with gather so autograd supplies d/dx, d/doffset, d/dw."""
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    p = lambda v: (v, v) if isinstance(v, int) else tuple(v)
    (sh, sw), (ph, pw), (dh, dw) = p(stride), p(padding), p(dilation)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    ys = (torch.arange(Ho, dtype=x.dtype) * sh - ph).view(1, 1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * sw - pw).view(1, 1, 1, Wo)
    ki = (torch.arange(kh, dtype=x.dtype) * dh).repeat_interleave(kw).view(1, kh * kw, 1, 1)
    kj = (torch.arange(kw, dtype=x.dtype) * dw).repeat(kh).view(1, kh * kw, 1, 1)
    off = offset.view(B, kh * kw, 2, Ho, Wo)
    hi = ys + ki + off[:, :, 0]
    wi = xs + kj + off[:, :, 1]
    inside = (hi > -1) & (wi > -1) & (hi < H) & (wi < W)
    h0 = torch.floor(hi)
    w0 = torch.floor(wi)
    lh, lw = hi - h0, wi - w0
    flat = x.reshape(B, C, H * W)

    def tap(hh, ww, wt):
        ok = inside & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
        idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1)).long().view(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).view(B, C, kh * kw, Ho, Wo)
        return v * (wt * ok.to(x.dtype)).unsqueeze(1)

    col = tap(h0, w0, (1 - lh) * (1 - lw)) + tap(h0, w0 + 1, (1 - lh) * lw) + \
        tap(h0 + 1, w0, lh * (1 - lw)) + tap(h0 + 1, w0 + 1, lh * lw)
    if mask is not None:
        col = col * mask.view(B, 1, kh * kw, Ho, Wo)
    if groups == 1:
        return torch.einsum('bckhw,ock->bohw', col, weight.reshape(Co, C, kh * kw))
    cg, cog = C // groups, Co // groups
    out = torch.einsum('bgckhw,gock->bgohw', col.reshape(B, groups, cg, kh * kw, Ho, Wo),
                       weight.reshape(groups, cog, cg, kh * kw))
    return out.reshape(B, Co, Ho, Wo)
