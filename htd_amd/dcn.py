"""Deformable convolution layers of the path: `DeformConv2dPack` (registry name 'DCN', what
configs/htd/htd_resnet101_dcn_2x_mstrain.py:142 asks for) and `ModulatedDeformConv2dPack` ('DCNv2').
Interface of mmcv.ops.{DeformConv2d,DeformConv2dPack,ModulatedDeformConv2dPack} as the reference uses them
(backbones/resnet.py:186-194, zero-initialised `conv_offset` :608-612; stale wrappers
build/lib/mmdet/ops/dcn/deform_conv.py:257-300,385-…).  GPU only: the reference has no CPU deformable conv either
(deform_conv.py:45-46 raises NotImplementedError).
"""
import math
import os

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import capi, dense
from .registry import CONV_LAYERS

CL = torch.channels_last
_P, _S = capi.ptr, capi.current_stream_ptr
KEEP_COLUMNS = os.environ.get('HTD_DCN_KEEP_COLUMNS', '1') != '0'     # 0: re-sample in backward (saves memory)


class DeformConv2dFunction(Function):
    @staticmethod
    def forward(ctx, x, offset, mask, weight, stride, padding, dilation, deform_groups, bias=None, relu=False,
                groups=1):
        if not x.is_cuda:
            raise NotImplementedError('deform_conv2d: only GPU tensors are supported')
        if x.dim() != 4:
            raise ValueError(f'Expected 4D tensor as input, got {x.dim()}D tensor instead.')   # deform_conv.py:27-29
        x = x.contiguous(memory_format=CL)
        offset = offset.contiguous(memory_format=CL)
        mask = mask.contiguous(memory_format=CL) if mask is not None else None
        weight = weight.contiguous(memory_format=CL)
        B, C, H, W = x.shape
        Co, cg, kh, kw = weight.shape
        if cg * groups != C or (groups > 1 and Co != C):
            raise ValueError(f'deform_conv2d: weight {tuple(weight.shape)} does not fit {C} channels in {groups} groups')
        Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
        Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
        if offset.shape != (B, 2 * deform_groups * kh * kw, Ho, Wo):
            raise ValueError(f'offset shape {tuple(offset.shape)} does not match output {(B, Ho, Wo)}')
        M, K = B * Ho * Wo, kh * kw * C
        cols = torch.empty(M, K, device=x.device, dtype=x.dtype)
        capi.call('htd_deform_im2col', _P(x), _P(offset), _P(mask), _P(cols), B, H, W, C, kh, kw, stride, padding,
                  dilation, deform_groups, _S(), work=('byte', 4.0 * M * K * 2))
        bias = bias.contiguous() if bias is not None else None
        if groups > 1:                                # ResNeXt: grouped GEMM over the gathered columns
            geom = (C, groups, kh, kw, stride, padding, dilation)
            y = dense._gconv_fwd_raw(cols, dense._gconv_pack(weight, groups, False), bias, geom, relu, (B, H, W))
        else:
            y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=CL)
            capi.call('htd_conv2d_fwd', _P(cols), _P(weight), _P(bias), None, 0, 0, _P(y), 1, M, 1, K, Co, 1, 1, 1, 0, 1,
                      int(bool(relu)), None, _S(), work=('flop', 2.0 * M * K * Co))     # bias / ReLU in the GEMM epilogue
        # the gathered columns are kept for the weight gradient (9x the input: ~5 GB over the 30 layers of R101-DCN at
        # B = 4, ~20 GB for X101 -- nothing against 288 GB of HBM, and it saves re-sampling every layer in backward)
        keep_cols = ctx.needs_input_grad[3] and KEEP_COLUMNS
        ctx.save_for_backward(x, offset, mask, weight, y if relu else None, cols if keep_cols else None)
        ctx.cfg = (stride, padding, dilation, deform_groups, Ho, Wo, bias is not None, groups)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, offset, mask, weight, y, cols = ctx.saved_tensors
        stride, padding, dilation, dg, Ho, Wo, has_bias, groups = ctx.cfg
        geom = (x.size(1), groups, weight.size(2), weight.size(3), stride, padding, dilation)
        B, C, H, W = x.shape
        Co, _, kh, kw = weight.shape
        M, K = B * Ho * Wo, kh * kw * C
        gy = gy.contiguous(memory_format=CL)
        if y is not None:
            from .dense import _mask_raw
            gy = _mask_raw(gy, y)
        gx = goff = gmask = gw = gb = None
        want_b = has_bias and ctx.needs_input_grad[8]
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or (mask is not None and ctx.needs_input_grad[2]):
            if groups > 1:
                gcol = dense._gconv_dgrad_raw(gy, dense._gconv_pack(weight, groups, True), geom, x.shape, cols=True)
            else:
                wT = torch.empty(K * Co, device=gy.device, dtype=gy.dtype)
                capi.call('htd_conv2d_flip_weights', _P(weight), _P(wT), Co, 1, 1, K, _S())
                gcol = torch.empty(M, K, device=gy.device, dtype=gy.dtype)
                capi.call('htd_conv2d_bwd_data', _P(gy), _P(wT), None, None, _P(gcol), 1, M, 1, K, Co, 1, 1, 1, 0, 1, None,
                          _S(), work=('flop', 2.0 * M * K * Co))
            if ctx.needs_input_grad[0]:
                gx = torch.empty((B, C, H, W), device=gy.device, dtype=gy.dtype, memory_format=CL).zero_()
            goff = torch.empty_like(offset, memory_format=CL)
            gmask = torch.empty_like(mask, memory_format=CL) if mask is not None else None
            capi.call('htd_deform_col2im', _P(x), _P(offset), _P(mask), _P(gcol), _P(gx), _P(goff), _P(gmask), B, H, W,
                      C, kh, kw, stride, padding, dilation, dg, _S())
        if ctx.needs_input_grad[3]:
            if cols is None:
                cols = torch.empty(M, K, device=gy.device, dtype=gy.dtype)
                capi.call('htd_deform_im2col', _P(x), _P(offset), _P(mask), _P(cols), B, H, W, C, kh, kw, stride,
                          padding, dilation, dg, _S())
            if groups > 1:
                gw = dense._gconv_wgrad_raw(cols, gy, weight, geom, x.shape, cols=True)
                gb = gy.sum((0, 2, 3)) if want_b else None
            else:
                gw = torch.empty((Co, C, kh, kw), device=gy.device, dtype=gy.dtype, memory_format=CL)
                nbytes = capi.lib().htd_conv2d_wgrad_workspace_bytes(1, M, 1, K, Co, 1, 1, 1, 0, 1)
                ws = torch.empty(nbytes // 4 + 1, device=gy.device, dtype=gy.dtype)
                gb = torch.empty(Co, device=gy.device, dtype=gy.dtype) if want_b else None
                capi.call('htd_conv2d_bwd_weight', _P(cols), _P(gy), _P(gw), _P(gb), 1, M, 1, K, Co, 1, 1, 1, 0, 1, _P(ws),
                          _S(), work=('flop', 2.0 * M * K * Co))
        elif want_b:
            gb = gy.sum((0, 2, 3))
        return gx, goff, gmask, gw, None, None, None, None, gb, None, None


class DeformConv2dBf16Function(Function):
    """Mixed-precision deformable convolution (bf16 mode, BASELINE configs[3]): bf16 activations / columns / gradient
    columns, fp32 offsets, fp32 master weight and bias.  The three GEMM halves run on the bf16 MFMA kernels
    (htd_conv2d_fwd_bf16 as a 1x1 over the column matrix, htd_conv2d_bwd_weight_bf16), sampling and scatter on the bf16
    forms of the im2col / col2im kernels; all accumulation is fp32."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, stride, padding, dilation, deform_groups, bias, relu):
        BF = torch.bfloat16
        x = x.contiguous(memory_format=CL)
        offset = offset.float().contiguous(memory_format=CL)
        mask = mask.float().contiguous(memory_format=CL) if mask is not None else None
        B, C, H, W = x.shape
        Co, _, kh, kw = weight.shape
        Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
        Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
        if offset.shape != (B, 2 * deform_groups * kh * kw, Ho, Wo):
            raise ValueError(f'offset shape {tuple(offset.shape)} does not match output {(B, Ho, Wo)}')
        M, K = B * Ho * Wo, kh * kw * C
        cols = torch.empty(M, K, device=x.device, dtype=BF)
        capi.call('htd_deform_im2col_bf16', _P(x), _P(offset), _P(mask), _P(cols), B, H, W, C, kh, kw, stride, padding,
                  dilation, deform_groups, _S(), work=('byte', 2.0 * M * K * 2))
        wb = weight.to(BF).contiguous(memory_format=CL)                 # [Co][kh][kw][C] = [Co][K]
        bias = bias.float().contiguous() if bias is not None else None
        y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=BF, memory_format=CL)
        capi.call('htd_conv2d_fwd_bf16', _P(cols), _P(wb), _P(bias), None, _P(y), 1, M, 1, K, Co, 1, 1, 1, 0, 1,
                  int(bool(relu)), _S(), work=('flop', 2.0 * M * K * Co))
        ctx.save_for_backward(x, offset, mask, wb, y if relu else None, cols if ctx.needs_input_grad[3] else None)
        ctx.cfg = (stride, padding, dilation, deform_groups, Ho, Wo, bias is not None, tuple(weight.shape))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        BF = torch.bfloat16
        x, offset, mask, wb, y, cols = ctx.saved_tensors
        stride, padding, dilation, dg, Ho, Wo, has_bias, wshape = ctx.cfg
        B, C, H, W = x.shape
        Co, _, kh, kw = wshape
        M, K = B * Ho * Wo, kh * kw * C
        gy = gy.to(BF).contiguous(memory_format=CL)
        if y is not None:
            gy = torch.ops.aten.threshold_backward(gy, y, 0)
        gx = goff = gmask = gw = gb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or (mask is not None and ctx.needs_input_grad[2]):
            wT = wb.permute(0, 2, 3, 1).reshape(Co, K).t().contiguous()   # [K = (tap, c)][Co]: gcol = gy @ W, 1x1 over M
            gcol = torch.empty(M, K, device=gy.device, dtype=BF)
            capi.call('htd_conv2d_fwd_bf16', _P(gy), _P(wT), None, None, _P(gcol), 1, M, 1, Co, K, 1, 1, 1, 0, 1, 0, _S(),
                      work=('flop', 2.0 * M * K * Co))
            gx32 = torch.empty((B, C, H, W), device=gy.device, dtype=torch.float32, memory_format=CL).zero_() \
                if ctx.needs_input_grad[0] else None
            goff = torch.empty_like(offset, memory_format=CL)
            gmask = torch.empty_like(mask, memory_format=CL) if mask is not None else None
            capi.call('htd_deform_col2im_bf16', _P(x), _P(offset), _P(mask), _P(gcol), _P(gx32), _P(goff), _P(gmask), B, H,
                      W, C, kh, kw, stride, padding, dilation, dg, _S())
            gx = gx32.to(BF) if gx32 is not None else None
        if ctx.needs_input_grad[3]:
            if cols is None:
                cols = torch.empty(M, K, device=gy.device, dtype=BF)
                capi.call('htd_deform_im2col_bf16', _P(x), _P(offset), _P(mask), _P(cols), B, H, W, C, kh, kw, stride,
                          padding, dilation, dg, _S())
            gw = dense.conv2d_wgrad_bf16(cols.view(1, M, 1, K).permute(0, 3, 1, 2), gy.permute(0, 2, 3, 1).reshape(
                1, M, 1, Co).permute(0, 3, 1, 2), (Co, K, 1, 1)).reshape(Co, kh, kw, C).permute(0, 3, 1, 2)
        if has_bias and ctx.needs_input_grad[8]:
            gb = torch.sum(gy.permute(0, 2, 3, 1).reshape(-1, Co), dim=0, dtype=torch.float32)
        return gx, goff, gmask, gw, None, None, None, None, gb, None


def deform_conv2d(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deform_groups=1, mask=None,
                  bias=None, relu=False):
    """mmcv.ops.deform_conv2d; bias / relu (extensions): folded-BN bias and ReLU in the epilogue of the GEMM half."""
    s, p, d = _pair(stride)[0], _pair(padding)[0], _pair(dilation)[0]
    if x.dtype == torch.bfloat16 and groups == 1 and x.size(1) % 32 == 0 and weight.size(0) % 32 == 0:
        return DeformConv2dBf16Function.apply(x, offset, mask, weight, s, p, d, deform_groups, bias, relu)
    return DeformConv2dFunction.apply(x, offset, mask, weight, s, p, d, deform_groups, bias, relu, int(groups))


class DeformConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=False, deformable_groups=None):
        super().__init__()
        assert not bias, 'mmcv DeformConv2d has no bias'
        if deformable_groups is not None:
            deform_groups = deformable_groups
        assert in_channels % groups == 0 and out_channels % groups == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = _pair(kernel_size), _pair(stride), \
            _pair(padding), _pair(dilation)
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size)
                                   .contiguous(memory_format=CL))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self.weight.data = self.weight.data.contiguous(memory_format=CL)
        return self

    def forward(self, x, offset):
        return deform_conv2d(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                             self.deform_groups)


@CONV_LAYERS.register_module('DCN')
class DeformConv2dPack(DeformConv2d):
    """Offsets predicted by a zero-initialised 3x3 conv of the same geometry (C -> deform_groups*2*kh*kw)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from .detector.bricks import Conv2d
        self.conv_offset = Conv2d(self.in_channels, self.deform_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                  kernel_size=self.kernel_size, stride=self.stride, padding=self.padding,
                                  dilation=self.dilation, bias=True)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x, relu=False, weight=None, bias=None):
        """weight / bias / relu: the frozen-BN-folded pair and the activation that follow this layer in a bottleneck
        (the offsets are always predicted from the layer's own parameters)."""
        offset = self.conv_offset(x.float() if x.dtype != torch.float32 else x)      # offsets are always fp32
        return deform_conv2d(x, offset, self.weight if weight is None else weight, self.stride, self.padding,
                             self.dilation, self.groups, self.deform_groups, bias=bias, relu=relu)


@CONV_LAYERS.register_module('DCNv2')
class ModulatedDeformConv2dPack(DeformConv2d):
    """DCNv2: conv_offset predicts 2*K offsets + K mask logits; mask = sigmoid."""

    def __init__(self, *args, bias=True, **kwargs):
        super().__init__(*args, bias=False, **kwargs)
        from .detector.bricks import Conv2d
        self.bias = nn.Parameter(torch.zeros(self.out_channels)) if bias else None
        taps = self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset = Conv2d(self.in_channels, self.deform_groups * 3 * taps, kernel_size=self.kernel_size,
                                  stride=self.stride, padding=self.padding, dilation=self.dilation, bias=True)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        out = self.conv_offset(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        y = deform_conv2d(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                          self.deform_groups, mask=torch.sigmoid(mask))
        return y if self.bias is None else y + self.bias.view(1, -1, 1, 1)
