"""FPN neck (mmdet/models/necks/fpn.py:9-216): 1x1 laterals, nearest top-down add, 3x3 output convs,
P6 = stride-2 subsample of P5.  The HTD configs use the plain variant (no extra convs, no norm)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..registry import NECKS
from .bricks import ConvModule, xavier_init


FUSED_TOP_DOWN = os.environ.get('HTD_FPN_FUSED', '1') != '0'      # 0: interpolate + add (A/B runs)


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest')):
        super().__init__()
        assert isinstance(in_channels, list)
        assert not add_extra_convs, 'extra FPN convs are outside the HTD path (P6 is a subsample of P5)'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.upsample_cfg = dict(upsample_cfg)
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels) and num_outs == end_level - start_level
        self.start_level, self.end_level, self.add_extra_convs = start_level, end_level, add_extra_convs
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg if not no_norm_on_lateral else None,
                                                 act_cfg=act_cfg, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    def _fused_top_down(self, inputs):
        """Plain lateral convs (no norm / activation) on GPU tensors, nearest up-sampling to the finer level's size."""
        if not inputs[0].is_cuda or inputs[0].dtype not in (torch.float32, torch.bfloat16) or \
                any(t.dtype != inputs[0].dtype for t in inputs) or \
                self.upsample_cfg.get('mode', 'nearest') != 'nearest' or \
                self.out_channels % 4 != 0 or not getattr(self, 'fused_top_down', FUSED_TOP_DOWN):
            return False
        if 'scale_factor' in self.upsample_cfg:      # must land exactly on the finer level's size
            sf = self.upsample_cfg['scale_factor']
            shapes = [t.shape[2:] for t in inputs[self.start_level:self.backbone_end_level]]
            if any(int(b[0] * sf) != a[0] or int(b[1] * sf) != a[1] for a, b in zip(shapes[:-1], shapes[1:])):
                return False
        return all(not m.with_norm and not m.with_activation for m in self.lateral_convs)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        n = len(self.lateral_convs)
        if self._fused_top_down(inputs):
            # top-down pathway (fpn.py:176-189) inside the lateral convolutions: level i-1's 1x1 conv adds the
            # nearest-up-sampled level i in its epilogue (no up-sampled map, no separate add)
            # Each coarser level i + 1 has two readers, its output convolution and level i's lateral sum: the output
            # convolution runs first and hands back an alias of its input (Conv2dFunction chain), the lateral reads the
            # alias, so in backward the summed-down gradient joins in the output convolution's data-gradient epilogue.
            laterals, outs_chain = [None] * n, [None] * n
            chain = inputs[0].is_cuda and inputs[0].dtype == torch.float32 and torch.is_grad_enabled() and \
                all(not c.with_norm and not c.with_activation and getattr(c, 'compute_dtype', None) is None
                    for c in self.fpn_convs[:n])
            for i in range(n - 1, -1, -1):
                m = self.lateral_convs[i]
                up = laterals[i + 1] if i + 1 < n else None
                laterals[i] = m.conv(inputs[i + self.start_level], residual=up, residual_up=up is not None)
                if chain and i > 0:
                    outs_chain[i], laterals[i] = self.fpn_convs[i].conv(laterals[i], chain=True)
            if chain:
                outs = [outs_chain[i] if i > 0 else self.fpn_convs[0](laterals[0]) for i in range(n)]
                for _ in range(self.num_outs - n):
                    outs.append(outs[-1][:, :, ::2, ::2])
                return tuple(outs)
        else:
            laterals = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
            for i in range(n - 1, 0, -1):
                if 'scale_factor' in self.upsample_cfg:
                    up = F.interpolate(laterals[i], **self.upsample_cfg)
                else:
                    up = F.interpolate(laterals[i], size=laterals[i - 1].shape[2:], **self.upsample_cfg)
                laterals[i - 1] = laterals[i - 1] + up
        outs = [self.fpn_convs[i](laterals[i]) for i in range(n)]
        for _ in range(self.num_outs - n):
            outs.append(outs[-1][:, :, ::2, ::2])      # == F.max_pool2d(x, 1, stride=2), fpn.py:197-199
        return tuple(outs)
