"""Stage-2 head of HTD: PGraph (local spatial aggregation + global semantic interaction) on the
classification branch and the BA-consuming convolutional regression branch.

Reference: roi_heads/bbox_heads/htd_bbox_head.py:22-230 (HTDBBoxHead).  Same registry name, kwargs
(including the misspelt `relpace`), state_dict keys and forward signature
    forward(x_cls, x_reg, feat, rois, fc_cls_0, enhanced_feat=None, pos_rois=None, global_feat=None).
Quirks kept on purpose: `fcs` runs twice (with and without global context), the
graph works on the branch WITHOUT it and the residual uses the branch WITH it; the global adjacency is a
soft-max over (1-M)*sim (local pairs get logit 0, not -inf); the prototype bank is stage 1's fc_cls
detached while fc_cls_0(x_cls) itself is live.
"""
import torch
import torch.nn as nn

from .. import dense
from .. import mmcv_ops as M
from ..registry import HEADS
from .bbox_heads import BBoxHead, TileLinear, fc_on_roi_tiles
from .bricks import ConvModule, normal_init, xavier_init
from .roi_extractors import map_roi_levels


@HEADS.register_module()
class HTDBBoxHead(BBoxHead):
    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=2, num_reg_convs=4,
                 num_reg_fcs=0, alpha=1, relpace=False, average=False, edge=1, conv_out_channels=256,
                 fc_out_channels=1024, conv_cfg=None, norm_cfg=dict(type='GN', num_groups=36), *args, **kwargs):
        kwargs.setdefault('with_avg_pool', True)
        super().__init__(*args, **kwargs)
        assert not relpace and not average, 'the HTD configs run with relpace=False, average=False'
        self.conv_kernel_size = 3
        self.num_shared_convs, self.num_shared_fcs = num_shared_convs, num_shared_fcs
        self.num_cls_convs, self.num_cls_fcs = num_cls_convs, num_cls_fcs
        self.num_reg_convs, self.num_reg_fcs = num_reg_convs, num_reg_fcs
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.alpha, self.relpace, self.edge, self.average = alpha, relpace, edge, average
        self.conv_out_channels = 1024
        self.fc_out_channels = 1024
        self.relu = nn.ReLU(inplace=True)
        self.gcn_in = self.gcn_out = 1024
        self.fc_cls = nn.Linear(self.fc_out_channels, self.num_classes + 1)
        self.fc_reg = nn.Linear(self.conv_out_channels, 4)
        self.middle_channel = 16 * 36
        convs = []
        for i in range(self.num_reg_convs):
            cin = self.in_channels if i == 0 else self.middle_channel
            last = i == self.num_reg_convs - 1
            convs.append(ConvModule(cin, 1024 if last else self.middle_channel, 3, stride=1, padding=1,
                                    conv_cfg=conv_cfg, norm_cfg=None if last else norm_cfg, bias=False))
        self.convs = nn.Sequential(*convs)
        fcs = []
        for i in range(self.num_cls_fcs):
            fcs.append(TileLinear(self.in_channels, self.roi_feat_size, self.fc_out_channels) if i == 0 else
                       nn.Linear(self.fc_out_channels, self.fc_out_channels))
            fcs.append(self.relu)
        self.fcs = nn.Sequential(*fcs)
        self.avg_pool = nn.AvgPool2d(self.roi_feat_size)
        self.graph_lvl0_cls = nn.Linear(self.gcn_in, self.gcn_out)
        self.graph_lvl1_cls = nn.Linear(self.gcn_in, self.gcn_out)
        self.graph_lvl2_cls = nn.Linear(self.gcn_in, self.gcn_out)
        self.graph_lvl3_cls = nn.Linear(self.gcn_in, self.gcn_out)
        self.graph_layer_cls = [self.graph_lvl0_cls, self.graph_lvl1_cls, self.graph_lvl2_cls, self.graph_lvl3_cls]

    def map_roi_levels(self, rois, num_levels):
        return map_roi_levels(rois, num_levels, 56)

    def init_weights(self):
        super().init_weights()
        normal_init(self.fc_cls, std=0.01)
        normal_init(self.fc_reg, std=0.001)
        for m in self.fcs.modules():
            if isinstance(m, nn.Linear):
                xavier_init(m, distribution='uniform')
        for m in self.graph_layer_cls:
            xavier_init(m, distribution='uniform')

    def _fuse_global(self, roi_feats, glbctx_feat, rois):
        assert roi_feats.size(0) == rois.size(0)
        return M.fuse_global(roi_feats, rois, glbctx_feat)

    def _cls_fcs(self, x):
        """fcs = Linear(12544,1024)+ReLU, Linear(1024,1024)+ReLU on NHWC RoI tiles."""
        lin = [m for m in self.fcs if isinstance(m, nn.Linear)]
        x = fc_on_roi_tiles(x, lin[0], relu=True, compute_dtype=getattr(self, 'compute_dtype', None))
        for fc in lin[1:]:
            x = dense.linear(x, fc.weight, fc.bias, relu=True)
        return x.float() if x.dtype != torch.float32 else x       # PGraph, classifier and losses: fp32

    def forward_reg(self, x_reg, enhanced_feat, pos_rois=None, global_feat=None):
        """Regression branch (htd_bbox_head.py:157-190): BA-enhanced positives -> 3 GN convs -> pool -> fc_reg."""
        if global_feat is not None:
            # x_reg + g[img] + alpha*enhanced in one pass (:163,184)
            x_reg = M.fuse_global(x_reg, pos_rois, global_feat, enhanced_feat, self.alpha)
        else:
            x_reg = x_reg + self.alpha * enhanced_feat
        x_reg = self.convs(x_reg)
        x_reg = M.global_avg_pool(x_reg).view(x_reg.size(0), x_reg.size(1))   # AvgPool2d(7) on a 7x7 map
        return dense.linear(x_reg, self.fc_reg.weight, self.fc_reg.bias) if self.with_reg else None

    def forward_cls(self, x_cls, feat, rois, fc_cls_0, global_feat=None, rois_per_img=None, roi_valid=None, row_stash=None):
        """Classification branch (:192-226): fcs (applied to the plain and to the global-fused tiles), semantic
        embedding from the stage-1 classifier, PGraph refinement, fc_cls."""
        from .pgraph import pgraph_refine
        prototype = torch.cat((fc_cls_0.weight, fc_cls_0.bias.unsqueeze(1)), 1).detach()
        if global_feat is not None:
            # the fcs run on the plain and on the global-fused tiles (:198,201): one batched pass over both
            tiles = M.plain_and_fused(x_cls, rois, global_feat, row_stash) if x_cls.is_cuda else \
                torch.cat([x_cls, self._fuse_global(x_cls, global_feat, rois)], 0)
            both = self._cls_fcs(tiles)
            x_cls, x_cls_glb = both[:x_cls.size(0)], both[x_cls.size(0):]
        else:
            x_cls, x_cls_glb = self._cls_fcs(x_cls), None
        # semantic embedding (:203-204): class posterior of the stage-1 classifier times its (detached) weights
        sam = dense.linear(dense.linear(x_cls, fc_cls_0.weight, fc_cls_0.bias).softmax(-1), prototype.t().contiguous())
        target_lvls = self.map_roi_levels(rois, len(feat))
        refined = pgraph_refine(x_cls, sam, rois, target_lvls, self.graph_layer_cls, rois_per_img, roi_valid)
        feat_cls_new = (x_cls_glb if global_feat is not None else x_cls) + refined
        return dense.linear(feat_cls_new, self.fc_cls.weight, self.fc_cls.bias) if self.with_cls else None

    def forward(self, x_cls, x_reg, feat, rois, fc_cls_0, enhanced_feat=None, pos_rois=None, global_feat=None,
                rois_per_img=None, roi_valid=None):
        bbox_pred = self.forward_reg(x_reg, enhanced_feat, pos_rois, global_feat)
        cls_score = self.forward_cls(x_cls, feat, rois, fc_cls_0, global_feat, rois_per_img, roi_valid)
        return cls_score, bbox_pred
