"""Operator surface of the HTD hot path on MI355X.

Mirrors the names, argument meaning and error behaviour of the `mmcv.ops` entry points
the reference's modules call (mmdet/ops/__init__.py:5-16): `RoIAlign`, `roi_align`,
`nms`, `batched_nms` (+ `soft_nms`, `DeformConv2dPack` in their own modules), and adds
the HTD-specific fused operators (`fuse_global`, `ba_fuse`, `global_avg_pool`,
`group_norm_relu`).  Every op calls libhtd_amd.so through the C ABI (htd_amd/capi.py);
tensors must live on the GPU -- a CPU tensor raises NotImplementedError exactly like the
reference wrappers (build/lib/mmdet/ops/roi_align/roi_align.py:39-40).  No fallbacks.

Memory layout contract: 4-D activations are logical NCHW tensors in
torch.channels_last memory format (= physical NHWC, what the kernels index).
"""
import ctypes
import os

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import capi

CL = torch.channels_last


def _drop_amax(t):
    """t is about to be modified in place by a kernel torch does not see: a maximum carried on it is void (dense.drop_amax)."""
    from . import dense
    dense.drop_amax(t)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise NotImplementedError(f'{name}: only GPU tensors are supported (libhtd_amd.so has no CPU path)')


def nhwc(x):
    """Logical NCHW -> channels_last memory (no copy if already so)."""
    return x.contiguous(memory_format=CL)


def _f32(x, name):
    if x.dtype != torch.float32:
        raise ValueError(f'{name}: expected float32, got {x.dtype}')
    return x


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


_S = capi.current_stream_ptr
_P = capi.ptr


# ====================================================================== RoIAlign
class PyramidTaps:
    """The FPN levels as seen by a SEQUENCE of RoIAlign consumers (stage-1 extractor, stage-2 extractor, BA).  Each
    consumer reads `levels`, and replaces them by identity aliases that its autograd node outputs; the next consumer
    therefore hangs off the previous one instead of off the pyramid itself.  In backward the gradient map of a level
    is then handed from node to node and every node scatter-adds into it IN PLACE: one zero fill and no
    `grad += other` per level instead of one of each per consumer."""

    def __init__(self, feats):
        self.levels = list(feats)

    def __len__(self):
        return len(self.levels)

    def __getitem__(self, i):
        return self.levels[i] if not isinstance(i, slice) else self.levels[i]


def _grad_buffer(galias, shape, device, dtype):
    """Gradient map to scatter into: the one handed down the tap chain when usable, else fresh zeros."""
    if galias is not None and galias.dtype == dtype and tuple(galias.shape) == tuple(shape) and \
            galias.is_contiguous(memory_format=CL):
        return galias
    buf = torch.empty(shape, device=device, dtype=dtype, memory_format=CL).zero_()
    if galias is not None:
        buf += galias
    return buf


# RoIAlign backward: 'gather' (default) -- a wavefront owns a strip of feature-map pixels and sums the RoIs covering it in
# RoI order: no float atomics, bit-stable, every pixel written once (no memset); 'scatter' -- the atomic kernels.
ROI_BWD = __import__('os').environ.get('HTD_ROI_BWD', 'gather')
ROI_BWD_ONE_LAUNCH = True        # gather form: all levels of a SingleRoIExtractor in one launch (False: one launch per level)


def _roi_align_bwd(g, rois, lvls, level, galias, shape, ph, pw, scale, sr, aligned):
    """-> gradient map of one level: galias (+)= RoIAlign^T(g), or a fresh map when nothing was handed down."""
    B, C, H, W = shape
    n = rois.size(0)
    if ROI_BWD != 'gather' or ph > 8 or pw > 8:
        gf = _grad_buffer(galias, shape, g.device, g.dtype)
        capi.call('htd_roi_align_bwd', _P(g), _P(rois), _P(lvls), level, _P(gf), n, B, C, H, W, ph, pw, scale, sr, aligned, _S())
        return gf
    usable = galias is not None and galias.dtype == g.dtype and tuple(galias.shape) == tuple(shape) and \
        galias.is_contiguous(memory_format=CL)
    if usable:
        gf, acc = galias, 1
        _drop_amax(gf)             # modified in place behind torch's back: a carried maximum (dense.tag_amax) is void
    else:
        gf = torch.empty(shape, device=g.device, dtype=g.dtype, memory_format=CL)
        acc = 0
        if galias is not None:
            gf.copy_(galias)
            acc = 1
    ws = torch.empty(capi.lib().htd_roi_align_bwd_gather_workspace_bytes(n), dtype=torch.uint8, device=g.device)
    # algorithmic bytes: the map written once (+ read when accumulating) + every RoI's 7x7xC gradient read once
    work = ('byte', 4.0 * (B * H * W * C * (1 + acc) + n * ph * pw * C)) if level in (0, None) or lvls is None else ('byte', 0.0)
    capi.call('htd_roi_align_bwd_gather', _P(g), _P(rois), _P(lvls), level if level is not None else 0, _P(gf), n, B, C, H, W,
              ph, pw, scale, sr, aligned, acc, _P(ws), _S(), work=work)
    return gf


ROI_FOLD = os.environ.get('HTD_ROI_FOLD', '1') != '0'                  # 0: every strip folds its RoIs' bins itself (A/B runs)
ROI_FOLD_MAX_BYTES = int(float(os.environ.get('HTD_ROI_FOLD_MAX_GB', '8')) * 2**30)
ROI_FOLD_MIN_ROIS = int(os.environ.get('HTD_ROI_FOLD_MIN_ROIS', '64'))
# One level per RoI (SingleRoIExtractor): a RoI is 7-14 pixels wide on ITS level, one or two strips per row, and the extra pass
# costs more than it saves (2048 RoIs: 470 -> 546 us).  BA pools every RoI from every level: ~20 strips per row on the fine ones.
ROI_FOLD_SINGLE = os.environ.get('HTD_ROI_FOLD_SINGLE', '0') != '0'


def _fold_workspace(n, Hs, L, pw, C, dev):
    """Folded-bin buffer of the two-pass gather backward (htd_roi_align_*_bwd_gather_folded), or None when folding is off, the
    buffer would not fit the cap, or there are too few RoIs for the extra launch to pay."""
    if not ROI_FOLD or n < ROI_FOLD_MIN_ROIS:
        return None
    nbytes = capi.lib().htd_roi_align_fold_workspace_bytes(n, Hs, L, pw, C)
    if nbytes > ROI_FOLD_MAX_BYTES:
        return None
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


def _roi_align_levels_bwd(g, rois, lvls, handed, need, shapes, ph, pw, scales, sr, aligned):
    """Gradient maps of every level that needs one, ONE launch (htd_roi_align_levels_bwd_gather): handed[i] (+)= RoIAlign_i^T(g),
    or a fresh map where nothing was handed down."""
    L, n = len(shapes), rois.size(0)
    maps, accs = [], []
    for i, shape in enumerate(shapes):
        if not need[i]:
            maps.append(None)
            accs.append(0)
            continue
        ga = handed[i]
        usable = ga is not None and ga.dtype == g.dtype and tuple(ga.shape) == tuple(shape) and ga.is_contiguous(memory_format=CL)
        if usable:
            _drop_amax(ga)
            maps.append(ga)
            accs.append(1)
        else:
            gf = torch.empty(shape, device=g.device, dtype=g.dtype, memory_format=CL)
            if ga is not None:
                gf.copy_(ga)
            maps.append(gf)
            accs.append(1 if ga is not None else 0)
    ws = torch.empty(L * capi.lib().htd_roi_align_bwd_gather_workspace_bytes(n), dtype=torch.uint8, device=g.device)
    ptrs = (ctypes.c_void_p * L)(*[m.data_ptr() if m is not None else None for m in maps])
    Hs = (ctypes.c_int * L)(*[s_[2] for s_ in shapes])
    Ws = (ctypes.c_int * L)(*[s_[3] for s_ in shapes])
    sc = (ctypes.c_float * L)(*[float(v) for v in scales])
    ac = (ctypes.c_int * L)(*accs)
    B, C = shapes[0][0], shapes[0][1]
    # algorithmic bytes: every map written once (+ read when accumulating) + every RoI's 7x7xC gradient read once
    work = ('byte', 4.0 * (sum(s_[0] * s_[2] * s_[3] * C * (1 + a) for s_, a, m in zip(shapes, accs, maps) if m is not None) +
                           n * ph * pw * C))
    fold = _fold_workspace(n, Hs, L, pw, C, g.device) if (C % 4 == 0 and ROI_FOLD_SINGLE) else None
    if fold is not None:
        capi.call('htd_roi_align_levels_bwd_gather_folded', _P(g), _P(rois), _P(lvls), ptrs, Hs, Ws, sc, ac, L, n, B, C, ph, pw,
                  int(sr), int(bool(aligned)), _P(ws), _P(fold), _S(), work=work, key='htd_roi_align_levels_bwd_gather')
    else:
        capi.call('htd_roi_align_levels_bwd_gather', _P(g), _P(rois), _P(lvls), ptrs, Hs, Ws, sc, ac, L, n, B, C, ph, pw, int(sr),
                  int(bool(aligned)), _P(ws), _S(), work=work)
    return maps


class RoIAlignFunction(Function):
    """mmcv.ops.roi_align semantics (avg pooling, aligned flag, sampling_ratio=0 => adaptive).  chain=True also
    returns an identity alias of `feat` (see PyramidTaps)."""

    @staticmethod
    def forward(ctx, feat, rois, output_size, spatial_scale, sampling_ratio, aligned, chain=False):
        _need_gpu(feat, 'roi_align')
        if rois.dim() != 2 or rois.size(1) != 5:
            raise AssertionError('RoI must be (idx, x1, y1, x2, y2)!')  # roi_align.py:136
        src = feat
        feat = nhwc(_f32(feat, 'roi_align'))
        rois = _f32(rois, 'roi_align').contiguous()
        ph, pw = _pair(output_size)
        B, C, H, W = feat.shape
        n = rois.size(0)
        out = torch.empty((n, C, ph, pw), device=feat.device, dtype=feat.dtype, memory_format=CL)
        capi.call('htd_roi_align_fwd', _P(feat), _P(rois), None, 0, _P(out), n, B, C, H, W, ph, pw,
                  float(spatial_scale), int(sampling_ratio), int(bool(aligned)), _S())
        ctx.save_for_backward(rois)
        ctx.args = ((B, C, H, W), ph, pw, float(spatial_scale), int(sampling_ratio), int(bool(aligned)))
        ctx.chain = bool(chain)
        ctx.set_materialize_grads(False)      # an unused alias must arrive as None, not as a full-size zero map
        return (out, src.view_as(src)) if chain else out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, galias=None):
        rois, = ctx.saved_tensors
        (B, C, H, W), ph, pw, scale, sr, aligned = ctx.args
        if not ctx.needs_input_grad[0] or (grad_out is None and galias is None):
            return (None, ) * 7
        if grad_out is None:
            return (galias, ) + (None, ) * 6
        grad_out = nhwc(grad_out)
        gfeat = _roi_align_bwd(grad_out, rois, None, 0, galias if ctx.chain else None, (B, C, H, W), ph, pw, scale, sr, aligned)
        return gfeat, None, None, None, None, None, None


def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
              chain=False):
    if pool_mode != 'avg':
        raise NotImplementedError("roi_align: only pool_mode='avg' (what the HTD configs use)")
    return RoIAlignFunction.apply(input, rois, output_size, spatial_scale, sampling_ratio, aligned, chain)


class RoIAlign(nn.Module):
    """Drop-in for mmcv.ops.RoIAlign as constructed by BaseRoIExtractor.build_roi_layers
    (roi_extractors/base_roi_extractor.py:49-56): RoIAlign(spatial_scale=1/s, output_size=7,
    sampling_ratio=0) -> aligned=True, avg pooling."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
                 use_torchvision=False):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.pool_mode = pool_mode
        self.aligned = aligned

    def forward(self, input, rois, chain=False):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.pool_mode,
                         self.aligned, chain)

    def __repr__(self):
        return (f'{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, '
                f'sampling_ratio={self.sampling_ratio}, pool_mode={self.pool_mode}, aligned={self.aligned})')


def roi_align_levels(feats, rois, target_lvls, output_size, scales, sampling_ratio=0, aligned=True):
    """All pyramid levels of SingleRoIExtractor.forward (single_level_roi_extractor.py:81-99) into one
    (N,C,ph,pw) tensor: level i's kernel only touches RoIs with target_lvls == i.  No nonzero(), no
    scatter, no host sync.  Differentiable w.r.t. every feats[i].  feats may be a PyramidTaps (chained gradients)."""
    if isinstance(feats, PyramidTaps):
        n = len(scales)
        res = _RoIAlignLevels.apply(rois, target_lvls, output_size, tuple(scales), sampling_ratio, aligned, True,
                                    *feats.levels[:n])
        feats.levels[:n] = list(res[1:])
        return res[0]
    return _RoIAlignLevels.apply(rois, target_lvls, output_size, tuple(scales), sampling_ratio, aligned, False, *feats)


class _RoIAlignLevels(Function):
    @staticmethod
    def forward(ctx, rois, lvls, output_size, scales, sampling_ratio, aligned, chain, *feats):
        _need_gpu(feats[0], 'roi_align')
        srcs = feats
        ph, pw = _pair(output_size)
        rois = _f32(rois, 'roi_align').contiguous()
        lvls = lvls.to(torch.int64).contiguous()
        n, C = rois.size(0), feats[0].size(1)
        # every RoI is written by exactly one level's launch (map_roi_levels clamps to [0, L)): no zero fill
        out = torch.empty((n, C, ph, pw), device=rois.device, dtype=torch.float32, memory_format=CL)
        shapes, fs = [], []
        for f in feats:
            f = nhwc(_f32(f, 'roi_align'))
            fs.append(f)
            shapes.append((f.size(0), C, f.size(2), f.size(3)))
        L = len(fs)
        if n:
            # one launch for all levels; algorithmic bytes: each RoI is pooled on ONE level: write ph*pw*C*4 B,
            # read its footprint, mid-range 21x21 px of the 14..28 px the level mapping yields
            ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in fs])
            Hs = (ctypes.c_int * L)(*[s_[2] for s_ in shapes])
            Ws = (ctypes.c_int * L)(*[s_[3] for s_ in shapes])
            sc = (ctypes.c_float * L)(*[float(v) for v in scales])
            work = ('byte', n * C * 4.0 * (ph * pw + 21 * 21))
            from . import dense
            if dense.emits('roi'):
                # the tiles go into the head's first FC layer: their maximum rides along for its H2 launches (dense.carried_amax)
                slot = dense._amax_slot(out.device)
                capi.call('htd_roi_align_levels_fwd_amax', ptrs, Hs, Ws, sc, L, _P(rois), _P(lvls), _P(out), n, shapes[0][0], C, ph,
                          pw, int(sampling_ratio), int(bool(aligned)), _P(slot), _S(), key='htd_roi_align_levels_fwd', work=work)
                dense.tag_amax(out, slot)
            else:
                capi.call('htd_roi_align_levels_fwd', ptrs, Hs, Ws, sc, L, _P(rois), _P(lvls), _P(out), n, shapes[0][0], C, ph, pw,
                          int(sampling_ratio), int(bool(aligned)), _S(), work=work)
        ctx.save_for_backward(rois, lvls)
        ctx.args = (shapes, ph, pw, scales, int(sampling_ratio), int(bool(aligned)))
        ctx.chain = bool(chain)
        ctx.set_materialize_grads(False)      # unused aliases must arrive as None, not as full-size zero maps
        return (out, *[f.view_as(f) for f in srcs]) if chain else out

    @staticmethod
    @once_differentiable
    def backward(ctx, g, *galias):
        rois, lvls = ctx.saved_tensors
        shapes, ph, pw, scales, sr, aligned = ctx.args
        if g is None:                            # pooled features unused: hand the chained maps on unchanged
            return (None, ) * 7 + tuple(galias[i] if (ctx.chain and i < len(galias)) else None for i in range(len(shapes)))
        g = nhwc(g)
        handed = [galias[i] if (ctx.chain and i < len(galias)) else None for i in range(len(shapes))]
        need = [bool(ctx.needs_input_grad[7 + i]) for i in range(len(shapes))]
        if ROI_BWD == 'gather' and ROI_BWD_ONE_LAUNCH and ph <= 8 and pw <= 8 and rois.size(0) > 0 and len(shapes) <= 6 and any(need):
            grads = _roi_align_levels_bwd(g, rois, lvls, handed, need, shapes, ph, pw, scales, sr, aligned)
            return (None, None, None, None, None, None, None, *grads)
        grads = []
        for i, (B, C, H, W) in enumerate(shapes):
            if not need[i]:
                grads.append(None)
                continue
            grads.append(_roi_align_bwd(g, rois, lvls, i, handed[i], (B, C, H, W), ph, pw, float(scales[i]), sr, aligned))
        return (None, None, None, None, None, None, None, *grads)


class _RoIAlignAllLevels(Function):
    """EVERY RoI pooled from EVERY level (AdptRoIExtractor / BA, adaptative_roi_extractor.py:66-76: one RoIAlign per level over
    the same RoI list) as ONE launch forward (htd_roi_align_all_levels_fwd) and ONE gather launch backward
    (htd_roi_align_all_levels_bwd_gather): launched level by level, the coarse maps' strips each walk all RoIs of their image
    while the rest of the chip idles.  -> (out_0 .. out_{L-1}[, alias_0 .. alias_{L-1}]); same values as L roi_align calls."""

    @staticmethod
    def forward(ctx, rois, output_size, scales, sampling_ratio, aligned, chain, *feats):
        _need_gpu(feats[0], 'roi_align')
        if rois.dim() != 2 or rois.size(1) != 5:
            raise AssertionError('RoI must be (idx, x1, y1, x2, y2)!')  # roi_align.py:136
        srcs = feats
        ph, pw = _pair(output_size)
        rois = _f32(rois, 'roi_align').contiguous()
        n, C, L = rois.size(0), feats[0].size(1), len(feats)
        fs = [nhwc(_f32(f, 'roi_align')) for f in feats]
        shapes = [(f.size(0), C, f.size(2), f.size(3)) for f in fs]
        outs = [torch.empty((n, C, ph, pw), device=rois.device, dtype=torch.float32, memory_format=CL) for _ in range(L)]
        if n:
            ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in fs])
            optr = (ctypes.c_void_p * L)(*[o.data_ptr() for o in outs])
            Hs = (ctypes.c_int * L)(*[s_[2] for s_ in shapes])
            Ws = (ctypes.c_int * L)(*[s_[3] for s_ in shapes])
            sc = (ctypes.c_float * L)(*[float(v) for v in scales])
            capi.call('htd_roi_align_all_levels_fwd', ptrs, Hs, Ws, sc, L, _P(rois), optr, n, shapes[0][0], C, ph, pw,
                      int(sampling_ratio), int(bool(aligned)), _S(), work=('byte', L * n * C * 4.0 * ph * pw))
        ctx.save_for_backward(rois)
        ctx.args = (shapes, ph, pw, tuple(float(v) for v in scales), int(sampling_ratio), int(bool(aligned)))
        ctx.chain = bool(chain)
        ctx.set_materialize_grads(False)      # unused outputs / aliases arrive as None, not as full-size zero maps
        return (*outs, *[f.view_as(f) for f in srcs]) if chain else tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        rois, = ctx.saved_tensors
        shapes, ph, pw, scales, sr, aligned = ctx.args
        L, n = len(shapes), rois.size(0)
        gouts = [nhwc(g) if g is not None else None for g in grads[:L]]
        handed = [grads[L + i] if (ctx.chain and L + i < len(grads)) else None for i in range(L)]
        need = [bool(ctx.needs_input_grad[6 + i]) for i in range(L)]
        dev = rois.device
        if n == 0 or ROI_BWD != 'gather' or ph > 8 or pw > 8 or L > 6:
            res = []
            for i, shape in enumerate(shapes):
                if not need[i] or gouts[i] is None:
                    res.append(handed[i] if need[i] else None)
                    continue
                res.append(_roi_align_bwd(gouts[i], rois, None, 0, handed[i], shape, ph, pw, scales[i], sr, aligned))
            return (None, ) * 6 + tuple(res)
        maps, accs = [], []
        for i, shape in enumerate(shapes):
            if not need[i] or gouts[i] is None:
                maps.append(None)
                accs.append(0)
                continue
            ga = handed[i]
            usable = ga is not None and ga.dtype == torch.float32 and tuple(ga.shape) == tuple(shape) and \
                ga.is_contiguous(memory_format=CL)
            if usable:
                _drop_amax(ga)
                maps.append(ga)
                accs.append(1)
            else:
                gf = torch.empty(shape, device=dev, dtype=torch.float32, memory_format=CL)
                if ga is not None:
                    gf.copy_(ga)
                maps.append(gf)
                accs.append(1 if ga is not None else 0)
        if any(m is not None for m in maps):
            ws = torch.empty(L * capi.lib().htd_roi_align_bwd_gather_workspace_bytes(n), dtype=torch.uint8, device=dev)
            gptr = (ctypes.c_void_p * L)(*[g.data_ptr() if (g is not None and m is not None) else None for g, m in zip(gouts, maps)])
            mptr = (ctypes.c_void_p * L)(*[m.data_ptr() if m is not None else None for m in maps])
            Hs = (ctypes.c_int * L)(*[s_[2] for s_ in shapes])
            Ws = (ctypes.c_int * L)(*[s_[3] for s_ in shapes])
            sc = (ctypes.c_float * L)(*scales)
            ac = (ctypes.c_int * L)(*accs)
            B, C = shapes[0][0], shapes[0][1]
            work = ('byte', 4.0 * sum(s_[0] * s_[2] * s_[3] * C * (1 + a) + n * ph * pw * C
                                      for s_, a, m in zip(shapes, accs, maps) if m is not None))
            fold = _fold_workspace(n, Hs, L, pw, C, dev) if C % 4 == 0 else None
            if fold is not None:
                capi.call('htd_roi_align_all_levels_bwd_gather_folded', gptr, _P(rois), mptr, Hs, Ws, sc, ac, L, n, B, C, ph, pw, sr,
                          aligned, _P(ws), _P(fold), _S(), work=work, key='htd_roi_align_all_levels_bwd_gather')
            else:
                capi.call('htd_roi_align_all_levels_bwd_gather', gptr, _P(rois), mptr, Hs, Ws, sc, ac, L, n, B, C, ph, pw, sr, aligned,
                          _P(ws), _S(), work=work)
        res = [m if m is not None else (handed[i] if need[i] else None) for i, m in enumerate(maps)]
        return (None, ) * 6 + tuple(res)


def roi_align_all_levels(feats, rois, output_size, scales, sampling_ratio=0, aligned=True):
    """[RoIAlign(feats[i], rois) for i] in one launch each way.  feats: list of maps or a PyramidTaps (chained gradients)."""
    L = len(scales)
    if isinstance(feats, PyramidTaps):
        res = _RoIAlignAllLevels.apply(rois, output_size, tuple(scales), sampling_ratio, aligned, True, *feats.levels[:L])
        feats.levels[:L] = list(res[L:])
        return list(res[:L])
    return list(_RoIAlignAllLevels.apply(rois, output_size, tuple(scales), sampling_ratio, aligned, False, *list(feats)[:L]))


# ====================================================================== max pooling (ResNet stem)
class MaxPool2dFunction(Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding):
        _need_gpu(x, 'max_pool2d')
        x = nhwc(_f32(x, 'max_pool2d'))
        B, C, H, W = x.shape
        Ho, Wo = (H + 2 * padding - kernel) // stride + 1, (W + 2 * padding - kernel) // stride + 1
        y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=CL)
        need = ctx.needs_input_grad[0]
        idx = torch.empty((B, Ho, Wo, C), device=x.device, dtype=torch.int32) if need else None
        from . import dense
        if dense.emits('pool'):                              # the 1x1 layers behind the pool read y on H2: its maximum rides along
            slot = dense._amax_slot(y.device)
            capi.call('htd_max_pool2d_fwd_amax', _P(x), _P(y), _P(idx), B, H, W, C, kernel, stride, padding, _P(slot), _S(),
                      key='htd_max_pool2d_fwd')
            dense.tag_amax(y, slot)
        else:
            capi.call('htd_max_pool2d_fwd', _P(x), _P(y), _P(idx), B, H, W, C, kernel, stride, padding, _S())
        ctx.save_for_backward(idx)
        ctx.args = ((B, C, H, W), kernel, stride, padding)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        (B, C, H, W), kernel, stride, padding = ctx.args
        g = nhwc(g)
        gx = torch.empty((B, C, H, W), device=g.device, dtype=g.dtype, memory_format=CL)
        capi.call('htd_max_pool2d_bwd', _P(g), _P(idx), _P(gx), B, H, W, C, kernel, stride, padding, _S())
        return gx, None, None, None


def max_pool2d(x, kernel_size, stride, padding=0):
    """nn.MaxPool2d(kernel_size, stride, padding) on NHWC maps (the ResNet stem pool)."""
    return MaxPool2dFunction.apply(x, int(kernel_size), int(stride), int(padding))


# ====================================================================== NMS
def nms_sorted_mask(sorted_boxes, iou_threshold, offset=0, seg_offsets=None, max_seg=None):
    """keep mask (uint8) for boxes already sorted by descending score.  With `seg_offsets`
    (int64 device tensor [S+1]) the rows form S independent problems handled in one launch."""
    _need_gpu(sorted_boxes, 'nms')
    b = _f32(sorted_boxes, 'nms').contiguous()
    n = b.size(0)
    keep = torch.zeros(n, dtype=torch.uint8, device=b.device)
    if n == 0:
        return keep
    if seg_offsets is None:
        ws = torch.empty(capi.lib().htd_nms_workspace_bytes(n), dtype=torch.uint8, device=b.device)
        capi.call('htd_nms_sorted', _P(b), _P(keep), n, float(iou_threshold), int(offset), _P(ws), _S())
    else:
        S = seg_offsets.numel() - 1
        max_seg = n if max_seg is None else int(max_seg)
        ncb = (max_seg + 63) // 64
        ws = torch.empty(n * ncb * 8 + 64, dtype=torch.uint8, device=b.device)
        seg64 = seg_offsets.to(torch.int64).contiguous()      # referenced until the launch is queued
        capi.call('htd_nms_sorted_batched', _P(b), _P(seg64), S, n, max_seg,
                  _P(keep), float(iou_threshold), int(offset), _P(ws), _S())
    return keep


def nms(boxes, scores, iou_threshold, offset=0):
    """mmcv.ops.nms: -> (dets (k,5), inds (k,) int64 in descending-score order).
    IoU > iou_threshold suppresses; ties in score keep the lower index first."""
    assert boxes.size(1) == 4 and boxes.size(0) == scores.size(0) and offset in (0, 1)
    _need_gpu(boxes, 'nms')
    order = torch.sort(scores, descending=True, stable=True)[1]
    keep = nms_sorted_mask(boxes[order], iou_threshold, offset)
    inds = order[keep.bool()]
    dets = torch.cat((boxes[inds], scores[inds].reshape(-1, 1)), dim=1)
    return dets, inds


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms (used at dense_heads/rpn_head.py:166-167 and
    core/post_processing/bbox_nms.py:65).  Boxes of different `idxs` never suppress each other.
    Like mmcv, IoUs are taken on boxes shifted by idx*(max_coordinate+1) -- the shift changes fp32
    rounding, so it is reproduced, not optimised away -- but each class is its own segment of one
    batched launch instead of relying on zero overlap (and there is no split_thr loop)."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    nms_type = cfg.pop('type', 'nms')
    cfg.pop('split_thr', None)
    if nms_type == 'soft_nms':
        from .soft_nms import soft_nms_batched
        return soft_nms_batched(boxes, scores, idxs, class_agnostic=class_agnostic, **cfg)
    if nms_type != 'nms':
        raise KeyError(f'unsupported nms type {nms_type}')
    thr = cfg.pop('iou_threshold', cfg.pop('iou_thr', None))
    offset = cfg.pop('offset', 0)
    _need_gpu(boxes, 'batched_nms')
    n = boxes.size(0)
    if n == 0:
        return boxes.new_zeros((0, 5)), boxes.new_zeros((0, ), dtype=torch.long)
    if class_agnostic:
        boxes_for_nms = boxes
        idxs = torch.zeros_like(idxs)
    else:
        max_coordinate = boxes.max()
        boxes_for_nms = boxes + (idxs.to(boxes) * (max_coordinate + 1))[:, None]
    order = torch.sort(scores, descending=True, stable=True)[1]          # global score order
    by_cls = torch.sort(idxs[order], stable=True)[1]                     # group by class, order kept inside
    perm = order[by_cls]
    cls_sorted = idxs[perm]
    counts = torch.bincount(cls_sorted)
    seg = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=boxes.device)
    seg[1:] = torch.cumsum(counts, 0)
    max_seg = int(counts.max().item())
    keep_sorted = nms_sorted_mask(boxes_for_nms[perm], thr, offset, seg, max_seg)
    keep_global = torch.zeros(n, dtype=torch.bool, device=boxes.device)
    keep_global[perm] = keep_sorted.bool()
    keep = order[keep_global[order]]
    return torch.cat([boxes[keep], scores[keep, None]], -1), keep


# ====================================================================== fuse_global
class FuseGlobalFunction(Function):
    """out = roi_feats + global_feat[img(roi)] (+ alpha * extra): HTDRoIHead._fuse_global
    htd_roi_head.py:133-141, and with `extra` the x_reg + g + alpha*enhanced of htd_bbox_head.py:163,184."""

    @staticmethod
    def forward(ctx, roi_feats, rois, global_feat, extra, alpha):
        _need_gpu(roi_feats, 'fuse_global')
        assert roi_feats.size(0) == rois.size(0)
        x = nhwc(_f32(roi_feats, 'fuse_global'))
        n, C, ph, pw = x.shape
        B = global_feat.size(0)
        g = global_feat.reshape(B, C).contiguous()
        e = nhwc(extra) if extra is not None else None
        rois = rois.contiguous()
        out = torch.empty_like(x, memory_format=CL)
        from . import dense
        if n and dense.emits('roi'):                         # the FC layer behind this reads `out` on H2: its maximum rides along
            slot = dense._amax_slot(out.device)
            capi.call('htd_fuse_global_fwd_amax', _P(x), _P(rois), _P(g), _P(e), float(alpha), _P(out), n, ph * pw, C, B, _P(slot), _S(),
                      key='htd_fuse_global_fwd')
            dense.tag_amax(out, slot)
        else:
            capi.call('htd_fuse_global_fwd', _P(x), _P(rois), _P(g), _P(e), float(alpha), _P(out), n, ph * pw, C, B, _S())
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(global_feat.shape), float(alpha), extra is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        rois, = ctx.saved_tensors
        gshape, alpha, has_extra = ctx.meta
        go = nhwc(go)
        n, C, ph, pw = go.shape
        gg = None
        if ctx.needs_input_grad[2]:
            gg = _fuse_global_grad(go, rois, n, ph * pw, C, gshape[0]).view(gshape)
        ge = go * alpha if (has_extra and ctx.needs_input_grad[3]) else None
        return go, None, gg, ge, None


def _fuse_global_grad(go, rois, n, P, C, B):
    """grad_global [B, C] = per-image sums of the RoI tile gradients: the bit-reproducible two-pass kernel (no float atomics)."""
    gg = torch.empty(B, C, device=go.device, dtype=go.dtype)
    if C % 4 == 0:
        ws = torch.empty((max(n, 1) + (max(n, 1) + 63) // 64 * B) * C, device=go.device, dtype=go.dtype)
        capi.call('htd_fuse_global_bwd_global_ws', _P(go), _P(rois), _P(gg), n, P, C, B, _P(ws), _S())
    else:
        gg.zero_()
        if n:
            capi.call('htd_fuse_global_bwd_global', _P(go), _P(rois), _P(gg), n, P, C, B, _S())
    return gg


class RowStash:
    """Side channel between PlainAndFusedFunction and select_rows_via: the gradient of the selected rows travels here instead
    of through a full-size dense tensor (see PlainAndFusedFunction)."""

    def __init__(self):
        self.alias = None        # identity alias of roi_feats, output of the PlainAndFused node
        self.rows = self.grad = None
        self.pending = False     # a select hangs off the alias and has not delivered its gradient yet


class PlainAndFusedFunction(Function):
    """(2n, C, ph, pw) = [roi_feats ; roi_feats + global_feat[img(roi)]]: the two inputs the HTD classification FCs run on
    (htd_bbox_head.py:198,201) as one batch, written by ONE kernel that reads roi_feats once (no fuse output, no torch.cat, no
    copy of the plain half); the gradient of roi_feats is ONE sum of the two halves.
    With a RowStash the node also outputs an identity alias of roi_feats for select_rows_via (the stage-2 positives of the
    regression branch, htd_roi_head.py:163-166, whose row indices are known only after this node has been queued): the
    gradient of those few rows is handed over through the stash and added into the sum IN PLACE.  As a second autograd
    consumer of roi_feats, index_select's backward (zero fill + index_add into default-strided storage) and the engine's
    accumulation cost a strided full-size add and a layout copy in the RoIAlign backward: 0.26 ms per step."""

    @staticmethod
    def forward(ctx, roi_feats, rois, global_feat, stash=None):
        _need_gpu(roi_feats, 'fuse_global')
        assert roi_feats.size(0) == rois.size(0)
        x = nhwc(_f32(roi_feats, 'fuse_global'))
        n, C, ph, pw = x.shape
        B = global_feat.size(0)
        g = global_feat.reshape(B, C).contiguous()
        rois = rois.contiguous()
        both = torch.empty((2 * n, C, ph, pw), device=x.device, dtype=x.dtype, memory_format=CL)
        from . import dense
        if n and dense.emits('roi'):
            slot = dense._amax_slot(both.device)
            capi.call('htd_plain_and_fused_fwd_amax', _P(x), _P(rois), _P(g), _P(both), n, ph * pw, C, B, _P(slot), _S(),
                      key='htd_plain_and_fused_fwd')
            dense.tag_amax(both, slot)
        else:
            capi.call('htd_plain_and_fused_fwd', _P(x), _P(rois), _P(g), _P(both), n, ph * pw, C, B, _S())
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(global_feat.shape), n, tuple(x.shape))
        ctx.stash = stash
        ctx.set_materialize_grads(False)
        if stash is None:
            return both
        return both, x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, go, galias=None):
        rois, = ctx.saved_tensors
        gshape, n, xshape = ctx.meta
        stash = ctx.stash
        gg = gx = None
        if go is not None:
            go = nhwc(go)
            C, ph, pw = go.shape[1:]
            if ctx.needs_input_grad[2]:
                gg = _fuse_global_grad(go[n:], rois, n, ph * pw, C, gshape[0]).view(gshape)
            if ctx.needs_input_grad[0]:
                gx = torch.add(go[:n], go[n:])
        if galias is not None and ctx.needs_input_grad[0]:       # someone else used the alias densely
            gx = nhwc(galias) if gx is None else gx.add_(galias)
        if stash is not None:
            if stash.pending:
                raise RuntimeError('PlainAndFused: the row selection hanging off its alias has not run its backward yet')
            if stash.grad is not None and ctx.needs_input_grad[0]:
                if gx is None:
                    gx = torch.zeros(xshape, device=stash.grad.device, dtype=stash.grad.dtype).contiguous(memory_format=CL)
                sg = stash.grad
                if gx.is_cuda and gx.dtype == torch.float32 and gx.is_contiguous(memory_format=CL) and sg.dtype == torch.float32 and \
                        (gx[0].numel() % 4) == 0:
                    sg = nhwc(sg)
                    _drop_amax(gx)                                # written behind torch's back
                    capi.call('htd_rows_add', _P(sg), _P(stash.rows.to(torch.int64).contiguous()), _P(gx), sg.size(0), gx.size(0),
                              gx[0].numel(), _S(), work=('byte', 12.0 * sg.numel()))
                else:
                    gx.index_add_(0, stash.rows, stash.grad)      # distinct rows: plain sums, deterministic
            stash.rows = stash.grad = None
        return gx, None, gg, None


class _SelectRowsVia(Function):
    @staticmethod
    def forward(ctx, alias, rows, stash):
        ctx.stash, ctx.rows = stash, rows
        stash.pending = True
        if alias.is_cuda and alias.dtype == torch.float32 and alias.dim() == 4 and alias.is_contiguous(memory_format=CL) and \
                (alias[0].numel() % 4) == 0:
            rows = rows.to(torch.int64).contiguous()
            out = torch.empty((rows.numel(), ) + tuple(alias.shape[1:]), device=alias.device, dtype=alias.dtype, memory_format=CL)
            capi.call('htd_rows_gather', _P(alias), _P(rows), _P(out), rows.numel(), alias.size(0), alias[0].numel(), _S(),
                      work=('byte', 8.0 * out.numel()))
            return out
        return torch.index_select(alias, 0, rows)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        ctx.stash.rows, ctx.stash.grad, ctx.stash.pending = ctx.rows, g, False
        return None, None, None               # delivered through the stash (PlainAndFusedFunction.backward)


def plain_and_fused(roi_feats, rois, global_feat, stash=None):
    """-> both; with a RowStash also stash.alias (see select_rows_via)"""
    if stash is None:
        return PlainAndFusedFunction.apply(roi_feats, rois, global_feat, None)
    both, stash.alias = PlainAndFusedFunction.apply(roi_feats, rois, global_feat, stash)
    return both


def select_rows_via(stash, rows):
    """roi_feats[rows] (rows: int64, no duplicates) read through the alias a plain_and_fused(..., stash) call left in the stash;
    differentiable, the gradient joins the PlainAndFused node's sum in place."""
    return _SelectRowsVia.apply(stash.alias, rows, stash)


def fuse_global(roi_feats, rois, global_feat, extra=None, alpha=1.0):
    return FuseGlobalFunction.apply(roi_feats, rois, global_feat, extra, alpha)


# ====================================================================== BA fusion
class BAFuseFunction(Function):
    """softmax-over-levels weighted sum of the per-level RoI features + P2 border ring
    (AdptRoIExtractor.forward adaptative_roi_extractor.py:76-91).  lvl_feats[0] is also the
    border source (roi_layers[0] on feats[0] is evaluated once, not twice)."""

    @staticmethod
    def forward(ctx, att, edge, *lvl_feats):
        _need_gpu(att, 'ba_fuse')
        L = len(lvl_feats)
        lv = [nhwc(_f32(f, 'ba_fuse')) for f in lvl_feats]
        n, C, ph, pw = lv[0].shape
        att = att.contiguous()
        assert att.shape == (L, n)
        out = torch.empty_like(lv[0], memory_format=CL)
        arr = (ctypes.c_void_p * L)(*[f.data_ptr() for f in lv])
        capi.call('htd_ba_fuse_fwd', arr, L, _P(lv[0]), _P(att), _P(out), n, ph, pw, C, int(edge), _S())
        ctx.save_for_backward(att, *lv)
        ctx.edge = int(edge)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        att, *lv = ctx.saved_tensors
        L = len(lv)
        n, C, ph, pw = lv[0].shape
        go = nhwc(go)
        glv = [torch.empty_like(f, memory_format=CL) for f in lv]
        gatt = torch.empty_like(att)
        arr = (ctypes.c_void_p * L)(*[f.data_ptr() for f in lv])
        garr = (ctypes.c_void_p * L)(*[f.data_ptr() for f in glv])
        # (no border map: the border source is level 0, the kernel adds the ring's gradient into glv[0])
        capi.call('htd_ba_fuse_bwd', arr, L, _P(att), _P(go), garr, None, _P(gatt), n, ph, pw, C, ctx.edge, _S())
        return (gatt, None, *glv)


def ba_fuse(att, lvl_feats, edge):
    return BAFuseFunction.apply(att, edge, *lvl_feats)


# ====================================================================== pooling / GN
class GlobalAvgPoolFunction(Function):
    """(n,C,h,w) -> (n,C,1,1) mean over h*w: nn.AdaptiveAvgPool2d(1) of SFA
    (global_context_head.py:372,386), BA attention (adaptative_roi_extractor.py:38) and the 7x7
    AvgPool of the reg branch (htd_bbox_head.py:122,188)."""

    @staticmethod
    def forward(ctx, x, chain=False):
        """chain=True: -> (pooled, identity alias of x).  A second consumer of x that reads the alias hands its gradient to
        THIS node's backward, which adds the pooling's share into it in place (htd_global_avg_pool_bwd_acc) -- x's producer
        gets one gradient map and autograd has nothing to add (BA: four (n,256,7,7) adds per step)."""
        _need_gpu(x, 'global_avg_pool')
        src = x
        x = nhwc(_f32(x, 'global_avg_pool'))
        n, C, h, w = x.shape
        out = torch.empty(n, C, device=x.device, dtype=x.dtype)
        capi.call('htd_global_avg_pool_fwd', _P(x), _P(out), n, h * w, C, _S())
        ctx.shape = (n, C, h, w)
        if chain:
            ctx.set_materialize_grads(False)
            return out.view(n, C, 1, 1), src.view_as(src)
        return out.view(n, C, 1, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, g, galias=None):
        n, C, h, w = ctx.shape
        if g is None:
            return galias, None
        g = g.reshape(n, C).contiguous()
        if galias is not None and galias.dtype == g.dtype and tuple(galias.shape) == (n, C, h, w) and \
                galias.is_contiguous(memory_format=CL):
            _drop_amax(galias)
            capi.call('htd_global_avg_pool_bwd_acc', _P(g), _P(galias), n, h * w, C, _S())
            return galias, None
        gx = torch.empty((n, C, h, w), device=g.device, dtype=g.dtype, memory_format=CL)
        capi.call('htd_global_avg_pool_bwd', _P(g), _P(gx), n, h * w, C, _S())
        return (gx if galias is None else gx + galias), None


def global_avg_pool(x, chain=False):
    return GlobalAvgPoolFunction.apply(x, chain)


class GroupNormReLUFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, num_groups, eps, relu):
        _need_gpu(x, 'group_norm')
        x = nhwc(_f32(x, 'group_norm'))
        n, C, h, w = x.shape
        y = torch.empty_like(x, memory_format=CL)
        mean = torch.empty(n, num_groups, device=x.device, dtype=x.dtype)
        rstd = torch.empty_like(mean)
        from . import dense
        slot = dense._amax_slot(x.device) if (n > 0 and capi.lib().htd_conv2d_set_h2(-1) == 1) else None
        if slot is not None:              # max |y| for the convolution behind this layer (H2 arithmetic, dense.carried_amax)
            capi.call('htd_group_norm_relu_fwd_amax', _P(x), _P(weight), _P(bias), _P(y), _P(mean), _P(rstd), n, h * w, C,
                      int(num_groups), float(eps), int(bool(relu)), _P(slot), _S())
            dense.tag_amax(y, slot)
        else:
            capi.call('htd_group_norm_relu_fwd', _P(x), _P(weight), _P(bias), _P(y), _P(mean), _P(rstd), n, h * w, C,
                      int(num_groups), float(eps), int(bool(relu)), _S())
        ctx.save_for_backward(x, y, weight, mean, rstd)
        ctx.meta = (int(num_groups), int(bool(relu)))
        ctx.bias_ref = bias                   # only its address is used (gradient sink lookup)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, y, weight, mean, rstd = ctx.saved_tensors
        G, relu = ctx.meta
        n, C, h, w = x.shape
        gy = nhwc(gy)
        gx = torch.empty_like(x, memory_format=CL)
        from . import dense
        gw = dense.grad_out(weight)           # straight into the flat gradient buffer when the parameters are registered there
        gb = dense.grad_out(ctx.bias_ref) if ctx.bias_ref is not None and ctx.bias_ref.shape == weight.shape else torch.empty_like(weight)
        ws = torch.empty(2 * max(n, 1) * C, device=x.device, dtype=torch.float32)      # per-tile sums, added in a fixed order
        L = capi.lib()
        if n > 0 and L.htd_conv2d_set_h2(-1) == 1 and L.htd_group_norm_bwd_amax_supported(h * w, C, G):
            slot = dense._amax_slot(x.device)
            capi.call('htd_group_norm_relu_bwd_amax', _P(x), _P(y), _P(weight), _P(mean), _P(rstd), _P(gy), _P(gx), _P(gw),
                      _P(gb), n, h * w, C, G, relu, _P(ws), _P(slot), _S(), key='htd_group_norm_relu_bwd_ws')
            dense.tag_amax(gx, slot)
        else:
            capi.call('htd_group_norm_relu_bwd_ws', _P(x), _P(y), _P(weight), _P(mean), _P(rstd), _P(gy), _P(gx), _P(gw),
                      _P(gb), n, h * w, C, G, relu, _P(ws), _S())
        return gx, gw, gb, None, None, None


def group_norm_relu(x, weight, bias, num_groups, eps=1e-5, relu=True):
    return GroupNormReLUFunction.apply(x, weight, bias, num_groups, eps, relu)


# ====================================================================== optimizer step
PARAM_EPOCH = 0


def sgd_momentum_step_(flat_param, flat_grad, flat_momentum, lr_dev, momentum, weight_decay, grad_scale=1.0):
    """In-place SGD(momentum, weight_decay) on flat fp32 buffers; lr_dev is a 1-element device tensor."""
    _need_gpu(flat_param, 'sgd')
    global PARAM_EPOCH
    PARAM_EPOCH += 1        # parameters change behind autograd's version counters: invalidates folded-weight caches
    capi.call('htd_sgd_momentum_step', _P(flat_param), _P(flat_grad), _P(flat_momentum), flat_param.numel(),
              _P(lr_dev), float(momentum), float(weight_decay), float(grad_scale), _S())


# ====================================================================== segmented top-k (RPN level ranking, samplers)
TOPK_CHUNK, TOPK_KMAX = 4096, 2048
_TOPK_PLANS = {}       # (segments, numel, device) -> device tables: built once per shape, never per call


def segmented_topk(keys, segments):
    """keys: contiguous float32 tensor; segments: sequence of (start, length, k) over keys.view(-1), 0 <= k <= min(length, 2048).
    -> (idx, val): for every segment, back to back, the positions (inside the segment) and values of its k largest keys in
    descending order, equal keys by ascending position (= `keys[start:start+length].sort(descending=True, stable=True)[:k]`)."""
    _need_gpu(keys, 'segmented_topk')
    keys = _f32(keys, 'segmented_topk')
    assert keys.is_contiguous()
    key = (tuple(segments), keys.numel(), str(keys.device))
    plan = _TOPK_PLANS.get(key)
    if plan is None:
        rows, chunks, out = [], [], 0
        for s, (start, length, k) in enumerate(segments):
            if not (0 <= k <= min(length, TOPK_KMAX)) or start < 0 or start + length > keys.numel():
                raise ValueError('segmented_topk: segment %d = (%d, %d, %d) out of range' % (s, start, length, k))
            rows.append((start, length, k, out))
            out += k
            chunks.extend((s, c) for c in range((length + TOPK_CHUNK - 1) // TOPK_CHUNK))
        if len(_TOPK_PLANS) > 256:
            _TOPK_PLANS.clear()
        plan = _TOPK_PLANS[key] = (len(rows), len(chunks), out,
                                   torch.tensor(rows, dtype=torch.int64, device=keys.device).view(-1, 4) if rows else None,
                                   torch.tensor(chunks, dtype=torch.int32, device=keys.device) if chunks else None)
    S, nchunks, out, segs, tab = plan
    idx = torch.empty(out, device=keys.device, dtype=torch.int64)
    val = torch.empty(out, device=keys.device, dtype=torch.float32)
    if S == 0 or out == 0:
        return idx, val
    ws = torch.empty(capi.lib().htd_segmented_topk_workspace_bytes(S, nchunks), dtype=torch.uint8, device=keys.device)
    capi.call('htd_segmented_topk', _P(keys), _P(segs), _P(tab) if tab is not None else None, S, nchunks, _P(idx), _P(val),
              _P(ws), _S())
    return idx, val
