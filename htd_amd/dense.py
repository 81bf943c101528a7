"""Dense contractions of the path (convolutions, Linear layers) -- the MFMA-bound part.

`conv2d` / `linear` are the single entry points every module uses.  They run the hand-written fp32
matrix-core implicit-GEMM kernels of libhtd_amd.so (htd_conv2d_fwd / _bwd_data / _bwd_weight,
csrc/conv_fwd.hip, csrc/conv_wgrad.hip) through the C ABI.  Activations NHWC, weights KRSC; the fused
epilogue carries bias, residual add and ReLU, and the backward fuses the ReLU mask with the bias gradient.
GPU tensors only -- there is no other path.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import capi

CL = torch.channels_last
_P = capi.ptr
_S = capi.current_stream_ptr


def _out_hw(H, W, kh, kw, stride, pad, dil):
    return ((H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1, (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise NotImplementedError(f'{name}: only GPU tensors are supported (libhtd_amd.so has no CPU path)')


def _splitk_ws(M, Co, Cred, kh, kw, device):
    """Split-K scratch for a GEMM with M rows, Co columns, reduction kh*kw*Cred (None when not worth splitting)."""
    nbytes = capi.lib().htd_conv2d_workspace_bytes(M, Co, Cred, kh, kw)
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32) if nbytes > 0 else None


class Conv2dFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, padding, dilation, relu):
        _need_gpu(x, 'conv2d')
        x = x.contiguous(memory_format=CL)
        weight = weight.contiguous(memory_format=CL)
        B, Ci, H, W = x.shape
        Co, Ci_w, kh, kw = weight.shape
        if Ci != Ci_w:
            raise ValueError(f'conv2d: input has {Ci} channels, weight expects {Ci_w}')
        Ho, Wo = _out_hw(H, W, kh, kw, stride, padding, dilation)
        y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=CL)
        res = residual.contiguous(memory_format=CL) if residual is not None else None
        b = bias.contiguous() if bias is not None else None
        flops = 2.0 * B * Ho * Wo * Co * kh * kw * Ci
        capi.call('htd_conv2d_fwd', _P(x), _P(weight), _P(b), _P(res), _P(y), B, H, W, Ci, Co, kh, kw, stride, padding,
                  dilation, int(bool(relu)), _P(_splitk_ws(B * Ho * Wo, Co, Ci, kh, kw, x.device)), _S(),
                  work=('flop', flops))
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.cfg = (stride, padding, dilation, bool(relu), bias is not None, residual is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        stride, padding, dilation, relu, has_bias, has_res = ctx.cfg
        B, Ci, H, W = x.shape
        Co, _, kh, kw = weight.shape
        Ho, Wo = g.shape[2], g.shape[3]
        g = g.contiguous(memory_format=CL)
        need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
        gb = None
        if relu or (has_bias and need_b):
            gm = torch.empty_like(g, memory_format=CL) if relu else g
            gb = torch.empty(Co, device=g.device, dtype=g.dtype)
            ws = torch.empty(2048 * Co, device=g.device, dtype=g.dtype)
            capi.call('htd_bias_grad_relu_mask', _P(g), _P(y) if relu else None, _P(gm) if relu else None, _P(gb),
                      B * Ho * Wo, Co, _P(ws), _S(), work=('byte', 4.0 * B * Ho * Wo * Co * (3 if relu else 1)))
            g = gm
        gx = gw = None
        flops = 2.0 * B * Ho * Wo * Co * kh * kw * Ci
        if need_x:
            gd, wd, Cod = g, weight, Co
            if Co % 8 != 0:      # skinny heads (RPN cls+reg, Co=15): pad the reduction channels with zeros
                padc = (-Co) % 8
                gd = torch.nn.functional.pad(g, (0, 0, 0, 0, 0, padc)).contiguous(memory_format=CL)
                wd = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0, 0, padc)).contiguous(memory_format=CL)
                Cod = Co + padc
            wT = torch.empty(Ci * kh * kw * Cod, device=g.device, dtype=g.dtype)
            capi.call('htd_conv2d_flip_weights', _P(wd), _P(wT), Cod, kh, kw, Ci, _S())
            gx = torch.empty((B, Ci, H, W), device=g.device, dtype=g.dtype, memory_format=CL)
            capi.call('htd_conv2d_bwd_data', _P(gd), _P(wT), None, _P(gx), B, H, W, Ci, Cod, kh, kw, stride, padding,
                      dilation, _P(_splitk_ws(B * H * W, Ci, Cod, kh, kw, g.device)), _S(), work=('flop', flops))
        if need_w:
            gw = torch.empty((Co, Ci, kh, kw), device=g.device, dtype=g.dtype, memory_format=CL)
            nbytes = capi.lib().htd_conv2d_wgrad_workspace_bytes(B, H, W, Ci, Co, kh, kw, stride, padding, dilation)
            ws = torch.empty(nbytes // 4 + 1, device=g.device, dtype=g.dtype)
            capi.call('htd_conv2d_bwd_weight', _P(x), _P(g), _P(gw), B, H, W, Ci, Co, kh, kw, stride, padding,
                      dilation, _P(ws), _S(), work=('flop', flops))
        return gx, gw, (gb if (has_bias and need_b) else None), (g if (has_res and need_r) else None), None, None, \
            None, None


def _pad_channels(x, weight, mult=8):
    """Zero-pad the channel dimension of (x, weight) to a multiple of `mult` (3-channel stem input)."""
    Ci = x.size(1)
    pad = (-Ci) % mult
    if pad == 0:
        return x, weight
    x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, pad))
    weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, pad))
    return x.contiguous(memory_format=CL), weight.contiguous(memory_format=CL)


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, residual=None):
    """y = act(conv2d(x, w) + bias + residual); x (B,Ci,H,W) channels_last, weight (Co,Ci,kh,kw) channels_last."""
    if isinstance(stride, (tuple, list)):
        stride, padding, dilation = stride[0], padding[0], dilation[0]
    if x.size(1) % 8 != 0:
        x, weight = _pad_channels(x, weight)
    return Conv2dFunction.apply(x, weight, bias, residual, int(stride), int(padding), int(dilation), relu)


def linear(x, weight, bias=None, relu=False):
    """y = act(x @ weight.T + bias); x (M,K), weight (N,K): the 1x1 convolution over M 'pixels'."""
    M, K = x.shape
    N = weight.size(0)
    if M == 0:
        return x.new_zeros(0, N) + (0 * weight.sum())
    if K % 8 != 0:
        pad = (-K) % 8
        x = torch.nn.functional.pad(x, (0, pad))
        weight = torch.nn.functional.pad(weight, (0, pad))
        K += pad
    x4 = x.contiguous().view(M, 1, 1, K).permute(0, 3, 1, 2)          # (M, K, 1, 1) channels_last view
    w4 = weight.contiguous().view(N, 1, 1, K).permute(0, 3, 1, 2)
    y = Conv2dFunction.apply(x4, w4, bias, None, 1, 0, 1, relu)       # (M, N, 1, 1) channels_last
    return y.permute(0, 2, 3, 1).reshape(M, N)


class BatchedGemmNT(Function):
    """c[g] = a[g] @ b[g]^T on the MFMA kernel (htd_bgemm_nt).  a (G,M,K), b (G,N,K) -> (G,M,N).
    Backward is two more NT products on transposed copies: ga = gc @ b, gb = gc^T @ a."""

    @staticmethod
    def _run(a, b):
        G, M, K = a.shape
        N = b.size(1)
        c = torch.empty(G, M, N, device=a.device, dtype=a.dtype)
        capi.call('htd_bgemm_nt', _P(a), _P(b), _P(c), G, M, N, K, _S(), work=('flop', 2.0 * G * M * N * K))
        return c

    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, 'bgemm_nt')
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return BatchedGemmNT._run(a, b)

    @staticmethod
    @once_differentiable
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = gc.contiguous()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = BatchedGemmNT._run(gc, b.transpose(1, 2).contiguous())                 # (G,M,N) x (G,K,N)^T
        if ctx.needs_input_grad[1]:
            gb = BatchedGemmNT._run(gc.transpose(1, 2).contiguous(), a.transpose(1, 2).contiguous())   # (G,N,M) x (G,K,M)^T
        return ga, gb


def bgemm_nt(a, b):
    """Batched a @ b^T; every dimension that becomes a row count must be a multiple of 128 when G > 1 and every
    reduction length a multiple of 8 (PGraph pads its groups accordingly)."""
    return BatchedGemmNT.apply(a, b)


def roofline_report(prof, peak_tflops, peak_gbs):
    """The `roofline` object of bench.py for the dominant hand-written kernel of the timed region:
    achieved = algorithmic work of its launches / their summed device time (live HIP-event timing)."""
    best = None
    for name, (calls, ms, kind, work) in prof.items():
        if kind is None or ms <= 0:
            continue
        if best is None or ms > best[2]:
            best = (name, calls, ms, kind, work)
    if best is None:
        return None
    name, calls, ms, kind, work = best
    if kind == 'flop':
        achieved = work / (ms * 1e-3) / 1e12
        return dict(kernel=name, bound='mfma', achieved=round(achieved, 3), peak=peak_tflops, unit='TFLOP/s',
                    frac=round(achieved / peak_tflops, 4), traffic=None, launches=calls,
                    avg_launch_ms=round(ms / calls, 4))
    achieved = work / (ms * 1e-3) / 1e9
    return dict(kernel=name, bound='hbm', achieved=round(achieved, 2), peak=peak_gbs, unit='GB/s',
                frac=round(achieved / peak_gbs, 4), traffic=None, launches=calls, avg_launch_ms=round(ms / calls, 4))
