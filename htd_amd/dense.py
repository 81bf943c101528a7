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
