"""ResNet-50/101 backbone of HTD (mmdet/models/backbones/resnet.py:95-300 Bottleneck, :303-649 ResNet,
mmdet/models/utils/res_layer.py:5-102).  Same constructor kwargs, module tree and state_dict keys
(`layer1.0.conv1.weight`, `layer1.0.bn1.running_mean`, `layer2.0.downsample.1.weight`, ...).

Execution differs from the reference: BatchNorm always runs with frozen statistics on this path
(norm_eval=True in every HTD config), so each conv->BN(->ReLU)(+identity) group is ONE fused
convolution: BN is folded into the conv weights/bias by differentiable tensor ops (bricks.frozen_bn_fold)
and ReLU / the residual add live in the conv epilogue.  Activations stay NHWC end to end.
"""
import math
import os

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from .. import mmcv_ops as M
from ..dense import ResStageBf16Function, ResStageFunction
from ..registry import BACKBONES
from .bricks import build_conv_layer, build_norm_layer, constant_init, frozen_bn_fold, frozen_bn_fold_many, kaiming_init

CL = torch.channels_last
FUSE_BF16_STAGE = os.environ.get('HTD_BF16_STAGE', '1') != '0'


def conv_bn(conv, bn, x, relu=False, residual=None):
    """conv -> BN (-> + residual) (-> ReLU) as one fused conv when BN uses its running statistics."""
    if bn.training:
        y = bn(conv(x))
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    w, b = frozen_bn_fold(conv.weight, bn)
    return conv(x, relu=relu, residual=residual, weight=w, bias=b)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', with_cp=False,
                 conv_cfg=None, norm_cfg=dict(type='BN'), dcn=None, plugins=None, groups=1, base_width=4,
                 base_channels=64):
        super().__init__()
        assert style in ['pytorch', 'caffe']
        assert dcn is None or isinstance(dcn, dict)
        assert plugins is None, 'plugins are outside the HTD path'
        self.inplanes, self.planes, self.stride, self.dilation = inplanes, planes, stride, dilation
        self.style, self.with_cp, self.conv_cfg, self.norm_cfg = style, with_cp, conv_cfg, norm_cfg
        self.dcn, self.with_dcn = dcn, dcn is not None
        self.conv1_stride, self.conv2_stride = (1, stride) if style == 'pytorch' else (stride, 1)
        # ResNeXt (backbones/resnext.py:33-38): the 3x3 runs `groups` groups of base_width * planes / base_channels
        self.groups = groups
        width = planes if groups == 1 else math.floor(planes * (base_width / base_channels)) * groups
        self.width = width
        gkw = dict(groups=groups) if groups != 1 else {}
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, width, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, width, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(norm_cfg, planes * self.expansion, postfix=3)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, width, kernel_size=1, stride=self.conv1_stride, bias=False)
        self.add_module(self.norm1_name, norm1)
        fallback_on_stride = False
        if self.with_dcn:
            dcn = dict(dcn)
            fallback_on_stride = dcn.pop('fallback_on_stride', False)
        if not self.with_dcn or fallback_on_stride:
            self.conv2 = build_conv_layer(conv_cfg, width, width, kernel_size=3, stride=self.conv2_stride,
                                          padding=dilation, dilation=dilation, bias=False, **gkw)
        else:
            assert conv_cfg is None, 'conv_cfg must be None for DCN'
            self.conv2 = build_conv_layer(dcn, width, width, kernel_size=3, stride=self.conv2_stride,
                                          padding=dilation, dilation=dilation, bias=False, **gkw)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = build_conv_layer(conv_cfg, width, planes * self.expansion, kernel_size=1, bias=False)
        self.add_module(self.norm3_name, norm3)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    @property
    def norm3(self):
        return getattr(self, self.norm3_name)

    def forward(self, x):
        out = conv_bn(self.conv1, self.norm1, x, relu=True)
        if hasattr(self.conv2, 'conv_offset') and (self.norm2.training or not out.is_cuda or
                                                   not getattr(self, 'fuse_dcn_bn', True) or
                                                   type(self.conv2).__name__ != 'DeformConv2dPack'):
            out = torch.relu(self.norm2(self.conv2(out)))      # deformable conv2 with live BN statistics / DCNv2
        elif hasattr(self.conv2, 'conv_offset'):     # deformable conv2: folded BN + ReLU in its GEMM epilogue
            w, b = frozen_bn_fold(self.conv2.weight, self.norm2)
            dt = out.dtype
            if dt == torch.bfloat16 and self.groups == 1:            # bf16 mode: bf16 sampling / GEMM kernels
                out = self.conv2(out, relu=True, weight=w, bias=b)
            else:                                    # fp32, or the grouped (ResNeXt) form whose kernels are fp32
                out = self.conv2(out.float() if dt != torch.float32 else out, relu=True, weight=w, bias=b)
                out = out.to(dt) if dt != torch.float32 else out
        else:
            out = conv_bn(self.conv2, self.norm2, out, relu=True)
        identity = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
        return conv_bn(self.conv3, self.norm3, out, relu=True, residual=identity)


class ResLayer(nn.Sequential):
    def __init__(self, block, inplanes, planes, num_blocks, stride=1, avg_down=False, conv_cfg=None,
                 norm_cfg=dict(type='BN'), downsample_first=True, **kwargs):
        assert not avg_down and downsample_first, 'ResNetV1d / hourglass variants are outside the HTD path'
        self.block = block
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                build_conv_layer(conv_cfg, inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                build_norm_layer(norm_cfg, planes * block.expansion)[1])
        layers = [block(inplanes=inplanes, planes=planes, stride=stride, downsample=downsample, conv_cfg=conv_cfg,
                        norm_cfg=norm_cfg, **kwargs)]
        inplanes = planes * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(inplanes=inplanes, planes=planes, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                **kwargs))
        super().__init__(*layers)

    def _fusable(self):
        """All blocks plain bottlenecks (no DCN), 'pytorch' style, BN on running statistics: the whole layer runs
        as one autograd node with the fused backward of dense.ResStageFunction."""
        for blk in self:
            if not isinstance(blk, Bottleneck) or blk.with_dcn or blk.style != 'pytorch' or blk.groups != 1:
                return False
            bns = [blk.norm1, blk.norm2, blk.norm3] + ([blk.downsample[1]] if blk.downsample is not None else [])
            if any(bn.training or not isinstance(bn, nn.BatchNorm2d) for bn in bns):
                return False
        return True

    def forward(self, x, chain=False):
        """chain=True (fp32 one-node stages only): -> (out, alias of x) -- see dense.ResStageFunction; otherwise (out, x)."""
        if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16) or not self._fusable() or \
                (x.dtype == torch.bfloat16 and not FUSE_BF16_STAGE):
            y = super().forward(x)              # CPU / DCN or grouped blocks: block by block
            return (y, x) if chain else y
        pairs = []
        for blk in self:
            pairs += [(blk.conv1.weight, blk.norm1), (blk.conv2.weight, blk.norm2), (blk.conv3.weight, blk.norm3)]
            if blk.downsample is not None:
                pairs.append((blk.downsample[0].weight, blk.downsample[1]))
        # every fold of the stage in one launch; for the fp32 stages it also leaves the flipped weight images for the data
        # gradients (the bf16 stages make their own bf16 operands from the folded weights)
        params = frozen_bn_fold_many(pairs, want_flips=x.dtype == torch.float32 and
                                     (x.requires_grad or any(w.requires_grad for w, _ in pairs)),
                                     want_planes=x.dtype == torch.float32)
        strides, has_ds = tuple(blk.conv2_stride for blk in self), tuple(blk.downsample is not None for blk in self)
        if x.dtype == torch.float32:
            # (a strided first block -- layers 2-4 of every ResNet -- cannot take the chained gradient: the strided data gradient
            #  of its 1x1 shortcut has no `accum` operand, htd_conv2d_bwd_data; autograd adds the two maps as before)
            use = bool(chain and x.requires_grad and torch.is_grad_enabled() and strides[0] == 1)
            y = ResStageFunction.apply(x, strides, self[0].dilation, has_ds, use, *params)
            if chain:
                return y if use else (y, x)
            return y
        y = ResStageBf16Function.apply(x, strides, self[0].dilation, has_ds, *params)
        return (y, x) if chain else y


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch',
                 deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False, zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        assert not deep_stem and not avg_down and plugins is None
        self.depth = depth
        stem_channels = base_channels if stem_channels is None else stem_channels
        self.stem_channels, self.base_channels, self.num_stages = stem_channels, base_channels, num_stages
        assert 1 <= num_stages <= 4
        self.strides, self.dilations = strides, dilations
        assert len(strides) == len(dilations) == num_stages
        self.out_indices = out_indices
        assert max(out_indices) < num_stages
        self.style, self.deep_stem, self.avg_down, self.frozen_stages = style, deep_stem, avg_down, frozen_stages
        self.conv_cfg, self.norm_cfg, self.with_cp, self.norm_eval = conv_cfg, norm_cfg, with_cp, norm_eval
        self.dcn, self.stage_with_dcn = dcn, stage_with_dcn
        if dcn is not None:
            assert len(stage_with_dcn) == num_stages
        self.zero_init_residual = zero_init_residual
        self.block, stage_blocks = self.arch_settings[depth]
        self.stage_blocks = stage_blocks[:num_stages]
        self.inplanes = stem_channels
        self.conv1 = build_conv_layer(conv_cfg, in_channels, stem_channels, kernel_size=7, stride=2, padding=3,
                                      bias=False)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, stem_channels, postfix=1)
        self.add_module(self.norm1_name, norm1)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = base_channels * 2**i
            res_layer = self.make_res_layer(block=self.block, inplanes=self.inplanes, planes=planes,
                                            num_blocks=num_blocks, stride=strides[i], dilation=dilations[i],
                                            style=style, avg_down=avg_down, with_cp=with_cp, conv_cfg=conv_cfg,
                                            norm_cfg=norm_cfg, dcn=self.dcn if self.stage_with_dcn[i] else None)
            self.inplanes = planes * self.block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, res_layer)
            self.res_layers.append(name)
        self._freeze_stages()
        self.feat_dim = self.block.expansion * base_channels * 2**(len(self.stage_blocks) - 1)

    def make_res_layer(self, **kwargs):
        return ResLayer(**kwargs)

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.norm1.eval()
            for m in [self.conv1, self.norm1]:
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
        elif pretrained is None:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    kaiming_init(m)
                elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                    constant_init(m, 1)
            if self.dcn is not None:
                for m in self.modules():
                    if isinstance(m, Bottleneck) and hasattr(m.conv2, 'conv_offset'):
                        constant_init(m.conv2.conv_offset, 0)
            if self.zero_init_residual:
                for m in self.modules():
                    if isinstance(m, Bottleneck):
                        constant_init(m.norm3, 0)
        else:
            raise TypeError('pretrained must be a str or None')

    def forward(self, x):
        x = x.contiguous(memory_format=CL)
        x = conv_bn(self.conv1, self.norm1, x, relu=True)
        mp = self.maxpool
        if x.is_cuda and x.size(1) % 4 == 0 and not mp.ceil_mode and mp.dilation == 1:
            x = M.max_pool2d(x, mp.kernel_size, mp.stride, mp.padding)      # NHWC kernel of libhtd_amd.so
        else:
            x = mp(x)
        if getattr(self, 'compute_dtype', torch.float32) == torch.bfloat16 and x.is_cuda:
            x = x.to(torch.bfloat16)            # bf16 configurations: stages (and the neck) run on the bf16 kernels
        outs = []
        for i, name in enumerate(self.res_layers):
            layer = getattr(self, name)
            if isinstance(layer, ResLayer) and outs and (i - 1) in self.out_indices:
                # the previous stage's output has two consumers, this stage and the neck: the neck reads the alias this
                # stage hands back, so its gradient joins inside the stage's backward (dense.ResStageFunction, chain)
                x, outs[-1] = layer(x, chain=True)
            else:
                x = layer(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self


@BACKBONES.register_module()
class ResNeXt(ResNet):
    """backbones/resnext.py:86-153: ResNet whose bottleneck 3x3 is grouped (X101-64x4d: groups=64, base_width=4).
    The grouped 3x3 -- plain or deformable -- runs on the slab kernels of csrc/gconv.hip."""

    def __init__(self, groups=1, base_width=4, **kwargs):
        self.groups, self.base_width = groups, base_width
        super().__init__(**kwargs)

    def make_res_layer(self, **kwargs):
        return ResLayer(groups=self.groups, base_width=self.base_width, base_channels=self.base_channels, **kwargs)
