"""Dense contractions of the path (convolutions, Linear layers) -- the MFMA-bound part.

`conv2d` / `linear` are the single entry points every module uses; they run on the fp32 MFMA
implicit-GEMM kernels of libhtd_amd.so (htd_conv2d_*).  Activations NHWC, weights KRSC.
"""
import torch
import torch.nn.functional as F

CL = torch.channels_last


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, residual=None):
    y = F.conv2d(x, weight, bias, stride, padding, dilation)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return y


def linear(x, weight, bias=None, relu=False):
    y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


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
