"""SFA -- Semantic Feature Aggregation (GlobalContextHead): global context vector from P6 plus an
image-level multi-label loss.  Reference: roi_heads/bbox_heads/global_context_head.py:323-401."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import dense
from .. import mmcv_ops as M
from ..registry import HEADS
from .bricks import ConvModule


@HEADS.register_module()
class GlobalContextHead(nn.Module):
    def __init__(self, num_ins, num_convs=4, in_channels=256, conv_out_channels=256, num_classes=81, loss_weight=1.0,
                 conv_cfg=None, norm_cfg=None, conv_to_res=False):
        super().__init__()
        assert not conv_to_res, 'conv_to_res is not used by HTDRoIHead (htd_roi_head.py:61-71)'
        self.num_ins, self.num_convs, self.in_channels = num_ins, num_convs, in_channels
        self.conv_out_channels, self.num_classes, self.loss_weight = conv_out_channels, num_classes, loss_weight
        self.conv_cfg, self.norm_cfg, self.conv_to_res, self.fp16_enabled = conv_cfg, norm_cfg, conv_to_res, False
        self.convs = nn.ModuleList()
        for i in range(self.num_convs):
            self.convs.append(ConvModule(self.in_channels if i == 0 else conv_out_channels, conv_out_channels, 3,
                                         padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg))
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(conv_out_channels, num_classes)
        self.criterion = nn.BCEWithLogitsLoss()

    def init_weights(self):
        nn.init.normal_(self.fc.weight, 0, 0.01)
        nn.init.constant_(self.fc.bias, 0)

    def forward(self, feats):
        x = feats[-1]
        for conv in self.convs:
            x = conv(x)
        x = M.global_avg_pool(x)                                  # (B, C, 1, 1)
        mc_pred = dense.linear(x.reshape(x.size(0), -1), self.fc.weight, self.fc.bias)
        return mc_pred, x

    def loss(self, pred, labels):
        targets = pred.new_zeros(pred.size())
        for i, label in enumerate(labels):                        # multi-hot of the image's gt classes
            targets[i].index_fill_(0, label, 1.0)
        return self.loss_weight * F.binary_cross_entropy_with_logits(pred, targets)
