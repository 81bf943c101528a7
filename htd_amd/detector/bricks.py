"""Layer bricks the reference gets from mmcv.cnn: `build_conv_layer`, `build_norm_layer`, `ConvModule`
and the init helpers (call sites: backbones/resnet.py:3-4,138-194, necks/fpn.py:3,69-87,
bbox_heads/htd_bbox_head.py:77-113, global_context_head.py:358-368).  Parameters keep the reference's
names and logical shapes (state_dict compatibility); conv weights are held in channels_last memory,
i.e. physically [Co][kh][kw][Ci], the layout the HIP kernels index.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import mmcv_ops as M
from ..registry import CONV_LAYERS

CL = torch.channels_last


def dense_conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, residual=None, residual_up=False,
                 chain=False):
    """y = act(conv2d(x, w) + bias + residual) on NHWC activations / KRSC weights.  bf16 activations take the
    mixed-precision kernels (bf16 operands, fp32 accumulate, fp32 master parameters and parameter gradients)."""
    from .. import dense
    if x.dtype == torch.bfloat16:
        y = dense.conv2d_bf16_autograd(x, weight, bias, stride, padding, dilation, relu, residual, residual_up)
        return (y, x) if chain else y                 # no chaining in the bf16 path: the alias is the tensor itself
    return dense.conv2d(x, weight, bias, stride, padding, dilation, relu, residual, residual_up, chain)


@CONV_LAYERS.register_module('Conv')
@CONV_LAYERS.register_module('Conv2d')
class Conv2d(nn.Conv2d):
    """nn.Conv2d with KRSC (channels_last) weight storage and the fused bias/ReLU/residual epilogue."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.padding_mode == 'zeros'
        assert self.groups == 1 or (self.in_channels == self.out_channels and self.bias is None), \
            'grouped convs: the in == out, bias-free 3x3 of the ResNeXt bottleneck'
        self.weight.data = self.weight.data.contiguous(memory_format=CL)

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self.weight.data = self.weight.data.contiguous(memory_format=CL)
        return self

    def forward(self, x, relu=False, residual=None, weight=None, bias=None, residual_up=False, chain=False):
        w = self.weight if weight is None else weight
        b = self.bias if bias is None else bias
        if self.groups > 1:
            from .. import dense
            assert residual is None and not chain
            dt = x.dtype                              # fp32 kernels: bf16 stages cast around the grouped 3x3
            y = dense.grouped_conv2d(x.float() if dt != torch.float32 else x, w, b, self.stride[0], self.padding[0],
                                     self.dilation[0], self.groups, relu)
            return y.to(dt) if dt != torch.float32 else y
        return dense_conv2d(x, w, b, self.stride[0], self.padding[0], self.dilation[0], relu, residual, residual_up, chain)


def build_conv_layer(cfg, *args, **kwargs):
    cfg_ = dict(type='Conv2d') if cfg is None else dict(cfg)
    layer_type = cfg_.pop('type')
    cls = CONV_LAYERS.get(layer_type)
    if cls is None:
        raise KeyError(f'Unrecognized conv type {layer_type}')
    return cls(*args, **kwargs, **cfg_)


def build_norm_layer(cfg, num_features, postfix=''):
    """-> (name, layer) with mmcv's abbreviations ('bn1', 'gn')."""
    cfg_ = dict(cfg)
    layer_type = cfg_.pop('type')
    requires_grad = cfg_.pop('requires_grad', True)
    cfg_.setdefault('eps', 1e-5)
    if layer_type in ('BN', 'BN2d'):
        layer, abbr = nn.BatchNorm2d(num_features, **cfg_), 'bn'
    elif layer_type == 'GN':
        assert 'num_groups' in cfg_
        layer, abbr = nn.GroupNorm(num_channels=num_features, **cfg_), 'gn'
    else:
        raise KeyError(f'Unrecognized norm type {layer_type}')
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


class _BNFold(torch.autograd.Function):
    """One launch forward, one backward (htd_bn_fold_fwd / _bwd) instead of ~15 element-wise kernels per conv."""

    @staticmethod
    def forward(ctx, w, gamma, beta, mean, var, eps):
        from .. import capi
        w = w.contiguous(memory_format=CL)
        Co = w.size(0)
        K = w.numel() // Co
        wf = torch.empty_like(w, memory_format=CL)
        bf = torch.empty(Co, device=w.device, dtype=w.dtype)
        capi.call('htd_bn_fold_fwd', capi.ptr(w), capi.ptr(gamma), capi.ptr(beta), capi.ptr(mean), capi.ptr(var),
                  float(eps), capi.ptr(wf), capi.ptr(bf), Co, K, capi.current_stream_ptr())
        ctx.save_for_backward(w, gamma, mean, var)
        ctx.eps = float(eps)
        ctx.beta_ref = beta                               # only its address is used (gradient sink lookup)
        from .. import dense
        dense.mark_side_consumed(wf)                      # its gradient is read by backward() below, on the side stream
        return wf, bf

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gwf, gbf):
        from .. import capi
        w, gamma, mean, var = ctx.saved_tensors
        Co = w.size(0)
        K = w.numel() // Co
        gwf = gwf.contiguous(memory_format=CL)
        gbf = gbf.contiguous()
        from .. import dense

        def launch():
            (gw, s1), (gg, s2), (gb, s3) = dense.grad_out2(w), dense.grad_out2(gamma), dense.grad_out2(ctx.beta_ref)
            capi.call('htd_bn_fold_bwd', capi.ptr(w), capi.ptr(gamma), capi.ptr(mean), capi.ptr(var), ctx.eps,
                      capi.ptr(gwf), capi.ptr(gbf), capi.ptr(gw), capi.ptr(gg), capi.ptr(gb), Co, K,
                      capi.current_stream_ptr())
            return (gw, gg, gb), s1 and s2 and s3
        if not dense.OVERLAP_WGRAD or (capi.profiling() and not dense._OVERLAP_IN_PROFILE):
            return launch()[0] + (None, None, None)
        # weight-gradient stream (dense.py): gwf was produced there; gbf comes from the main stream
        main, side = torch.cuda.current_stream(), dense.side_stream(gwf.device)
        side.wait_stream(main)
        gwf.record_stream(side)
        gbf.record_stream(side)
        with torch.cuda.stream(side):
            outs, all_sinks = launch()
        if not all_sinks:
            for t in outs:
                t.record_stream(main)
            main.wait_stream(side)
        return outs + (None, None, None)


class _BNFoldMany(torch.autograd.Function):
    """frozen_bn_fold of EVERY conv + BN pair of a ResNet stage: one launch forward (htd_bn_fold_many_fwd, which also
    writes the flipped / transposed images the data gradients take), one backward (htd_bn_fold_many_bwd), instead of
    one ~5-microsecond launch per pair and pass plus one weight flip per data gradient (R50: 52 + 42 + 52 launches per
    step).  Inputs per pair: w, gamma, beta, mean, var; outputs per pair: w', b'."""

    @staticmethod
    def forward(ctx, eps, want_flips, want_planes, *tensors):
        import numpy as np
        from .. import capi, dense
        n = len(tensors) // 5
        dev = tensors[0].device
        ws = [tensors[5 * i].contiguous(memory_format=CL) for i in range(n)]
        sizes = [(w.size(0), w.size(1), w.size(2) * w.size(3)) for w in ws]            # Co, Ci, taps
        total = sum(2 * w.numel() + w.size(0) for w in ws) if want_flips else sum(w.numel() + w.size(0) for w in ws)
        flat = torch.empty(total, device=dev, dtype=torch.float32)
        desc = np.zeros((n, 10), dtype=np.int64)
        outs, flips, off, tile0 = [], [], 0, 0
        for i, (w, (Co, Ci, taps)) in enumerate(zip(ws, sizes)):
            wf = flat[off:off + w.numel()].view(Co, w.size(2), w.size(3), Ci).permute(0, 3, 1, 2)      # KRSC memory
            off += w.numel()
            bf = flat[off:off + Co]
            off += Co
            wT = None
            if want_flips:
                wT = flat[off:off + w.numel()]
                off += w.numel()
            g, b, m, v = tensors[5 * i + 1:5 * i + 5]
            desc[i, :8] = (w.data_ptr(), g.data_ptr(), b.data_ptr(), m.data_ptr(), v.data_ptr(), wf.data_ptr(), bf.data_ptr(),
                           wT.data_ptr() if wT is not None else 0)
            desc[i, 8] = Co | (Ci << 32)                     # two int32 per int64 slot (little endian): Co, Ci
            desc[i, 9] = taps | (tile0 << 32)                # taps, tile0
            tile0 += taps * ((Co + 31) // 32) * ((Ci + 31) // 32)
            outs += [wf, bf]
            flips.append(wT)
        table = capi.upload_table(desc, dev)
        capi.call('htd_bn_fold_many_fwd', capi.ptr(table), n, tile0, float(eps), capi.current_stream_ptr())
        # bf16 plane images of the folded weights (conv_x3p_kernel operands, csrc/conv_x3.hip): one more launch per stage
        if want_planes:                                      # fp32 stages only: the bf16 stages make bf16 operands of their own
            dense.planes_many([(wf, False) for wf in outs[0::2]] + ([(wf, True) for wf in outs[0::2]] if want_flips else []))
        for wf, wT in zip(outs[0::2], flips):
            dense.mark_side_consumed(wf)
            if wT is not None:
                dense.offer_flipped(wf, wT)              # picked up by ResStageFunction.forward, see dense.py
        ctx.save_for_backward(*[t for i in range(n) for t in (ws[i], tensors[5 * i + 1], tensors[5 * i + 3], tensors[5 * i + 4])])
        ctx.beta_refs = [tensors[5 * i + 2] for i in range(n)]
        ctx.eps, ctx.n = float(eps), n
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        import numpy as np
        from .. import capi, dense
        n, saved = ctx.n, ctx.saved_tensors
        dev = saved[0].device

        def launch():
            desc = np.zeros((n, 11), dtype=np.int64)
            outs, row0, all_sinks, keep = [], 0, True, []
            for i in range(n):
                w, gamma, mean, var = saved[4 * i:4 * i + 4]
                Co, K = w.size(0), w.numel() // w.size(0)
                gwf, gbf = grads[2 * i], grads[2 * i + 1]
                gwf = torch.zeros_like(w, memory_format=CL) if gwf is None else gwf.contiguous(memory_format=CL)
                gbf = torch.zeros(Co, device=dev) if gbf is None else gbf.contiguous()
                keep += [gwf, gbf]
                (gw, s1), (gg, s2), (gb, s3) = dense.grad_out2(w), dense.grad_out2(gamma), dense.grad_out2(ctx.beta_refs[i])
                all_sinks = all_sinks and s1 and s2 and s3
                desc[i, :9] = (w.data_ptr(), gamma.data_ptr(), mean.data_ptr(), var.data_ptr(), gwf.data_ptr(), gbf.data_ptr(),
                               gw.data_ptr(), gg.data_ptr(), gb.data_ptr())
                desc[i, 9] = Co | (K << 32)
                desc[i, 10] = row0
                row0 += Co
                outs.append((gw, gg, gb))
            table = capi.upload_table(desc, dev)
            capi.call('htd_bn_fold_many_bwd', capi.ptr(table), n, row0, ctx.eps, capi.current_stream_ptr())
            return outs, all_sinks, keep
        if not dense.OVERLAP_WGRAD or (capi.profiling() and not dense._OVERLAP_IN_PROFILE):
            outs = launch()[0]
        else:       # the weight gradients were produced on the side stream (dense.py)
            main, side = torch.cuda.current_stream(), dense.side_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                outs, all_sinks, keep = launch()
            for t in keep:
                t.record_stream(side)
            if not all_sinks:
                for trio in outs:
                    for t in trio:
                        t.record_stream(main)
                main.wait_stream(side)
        res = [None, None, None]
        for gw, gg, gb in outs:
            res += [gw, gg, gb, None, None]
        return tuple(res)


def frozen_bn_fold_many(pairs, want_flips=True, want_planes=True):
    """[(conv_weight, bn), ...] -> [w'_0, b'_0, w'_1, b'_1, ...]: every fold of a stage in one launch (training, fp32
    weights with K % 4 == 0 on the GPU); otherwise pair by pair."""
    ok = torch.is_grad_enabled() and all(w.is_cuda and w.dtype == torch.float32 and (w.numel() // w.size(0)) % 4 == 0 and
                                          not bn.training for w, bn in pairs)
    if not ok or len({bn.eps for _, bn in pairs}) != 1:
        out = []
        for w, bn in pairs:
            out += frozen_bn_fold(w, bn)
        return out
    args = []
    for w, bn in pairs:
        args += [w, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    return list(_BNFoldMany.apply(pairs[0][1].eps, bool(want_flips), bool(want_planes), *args))


def frozen_bn_fold(conv_weight, bn):
    """Eval-mode BatchNorm (norm_eval=True, backbones/resnet.py:640-649) folded into the preceding conv:
    w' = w * s, b' = beta - mean * s with s = gamma / sqrt(var + eps).  The backward of the fold returns exactly
    d/dgamma, d/dbeta, d/dw of the unfused conv->BN pair (no conv output is kept for the BN backward)."""
    if conv_weight.is_cuda and (conv_weight.numel() // conv_weight.size(0)) % 4 == 0:
        if torch.is_grad_enabled():
            return _BNFold.apply(conv_weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        # inference: the folded pair is a constant of the parameters -- keep it until they change
        key = (M.PARAM_EPOCH, conv_weight.data_ptr(), conv_weight._version, bn.weight._version, bn.bias._version,
               bn.running_mean._version, bn.running_var._version)
        cached = bn.__dict__.get('_htd_fold_cache')
        if cached is None or cached[0] != key:
            cached = (key, _BNFold.apply(conv_weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps))
            bn.__dict__['_htd_fold_cache'] = cached
        return cached[1]
    s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)          # 3-channel stem (K = 147): plain tensor ops
    return conv_weight * s.view(-1, 1, 1, 1), bn.bias - bn.running_mean * s


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if distribution == 'uniform' and gain == 1 and hasattr(module, 'tile_shape'):
        module.xavier_uniform_()                 # TileLinear: fans of the logical 2-D matrix
    elif distribution == 'uniform':
        nn.init.xavier_uniform_(module.weight, gain=gain)
    else:
        nn.init.xavier_normal_(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


class ConvModule(nn.Module):
    """conv -> norm -> ReLU block (mmcv.cnn.ConvModule with order conv/norm/act, bias='auto').
    GroupNorm+ReLU runs as one fused HIP kernel; a ReLU with no norm is fused into the conv epilogue."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, **unused):
        super().__init__()
        assert act_cfg is None or act_cfg['type'] == 'ReLU'
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.with_bias = bias
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)
        self.init_weights()

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def init_weights(self):
        kaiming_init(self.conv, a=0, nonlinearity='relu')
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    def forward(self, x):
        # compute_dtype = torch.bfloat16 (set by configs.build_htd_detector(bf16=True)): the convolution runs on the bf16
        # MFMA kernels, the normalisation and everything downstream stay fp32
        low = getattr(self, 'compute_dtype', None) == torch.bfloat16 and x.is_cuda and x.dtype == torch.float32 and \
            self.conv.in_channels % 32 == 0 and self.conv.out_channels % 8 == 0 and self.conv.groups == 1
        if low:
            x = x.to(torch.bfloat16)
        if not self.with_norm:
            y = self.conv(x, relu=self.with_activation)
            return y.float() if low else y
        x = self.conv(x)
        if low:
            x = x.float()
        norm = self.norm
        if isinstance(norm, nn.GroupNorm):
            return M.group_norm_relu(x, norm.weight, norm.bias, norm.num_groups, norm.eps, self.with_activation)
        x = norm(x)
        return F.relu(x) if self.with_activation else x
