"""RoI feature extractors: SingleRoIExtractor and AdptRoIExtractor (BA, Border-aware Adaptation).

Reference: roi_extractors/base_roi_extractor.py:8-83, single_level_roi_extractor.py:8-99,
adaptative_roi_extractor.py:8-91.  Same registry names, kwargs, state_dict keys (AdptRoIExtractor keeps the
aliased `conv1`/`att.1`, `conv2`/`att.3` parameters of the reference module tree).
"""
import torch
import torch.nn as nn

from .. import mmcv_ops as M
from ..registry import ROI_EXTRACTORS

_ROI_LAYERS = {'RoIAlign': M.RoIAlign}


class BaseRoIExtractor(nn.Module):
    def __init__(self, roi_layer, out_channels, featmap_strides):
        super().__init__()
        self.roi_layers = self.build_roi_layers(roi_layer, featmap_strides)
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.fp16_enabled = False

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def build_roi_layers(self, layer_cfg, featmap_strides):
        cfg = dict(layer_cfg)
        layer_type = cfg.pop('type')
        assert layer_type in _ROI_LAYERS, f'{layer_type} is not an RoI layer of this path'
        return nn.ModuleList([_ROI_LAYERS[layer_type](spatial_scale=1 / s, **cfg) for s in featmap_strides])

    def roi_rescale(self, rois, scale_factor):
        cx, cy = (rois[:, 1] + rois[:, 3]) * 0.5, (rois[:, 2] + rois[:, 4]) * 0.5
        w, h = (rois[:, 3] - rois[:, 1]) * scale_factor, (rois[:, 4] - rois[:, 2]) * scale_factor
        return torch.stack((rois[:, 0], cx - w * 0.5, cy - h * 0.5, cx + w * 0.5, cy + h * 0.5), dim=-1)


def map_roi_levels(rois, num_levels, finest_scale=56):
    """scale < 2*finest -> 0, < 4*finest -> 1, ... (single_level_roi_extractor.py:32-51,
    htd_bbox_head.py:129-135)."""
    if rois.is_cuda and rois.dtype == torch.float32 and rois.dim() == 2 and rois.size(1) == 5 and not rois.requires_grad:
        from .. import capi
        r = rois.contiguous()
        lvls = torch.empty(r.size(0), device=r.device, dtype=torch.int64)
        capi.call('htd_map_roi_levels', capi.ptr(r), capi.ptr(lvls), r.size(0), int(num_levels), float(finest_scale),
                  capi.current_stream_ptr())
        return lvls
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lvls.clamp(min=0, max=num_levels - 1).long()


@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(BaseRoIExtractor):
    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super().__init__(roi_layer, out_channels, featmap_strides)
        self.finest_scale = finest_scale

    def map_roi_levels(self, rois, num_levels):
        return map_roi_levels(rois, num_levels, self.finest_scale)

    def forward(self, feats, rois, roi_scale_factor=None):
        """One (N,C,7,7) tensor filled by one level-filtered kernel per pyramid level: no nonzero(),
        no per-level gather/scatter, no host synchronisation (reference :81-99)."""
        l0 = self.roi_layers[0]
        if len(feats) == 1:
            if len(rois) == 0:
                return feats[0].new_zeros(0, self.out_channels, *l0.output_size)
            return l0(feats[0], rois)
        lvls = self.map_roi_levels(rois, len(feats))
        if roi_scale_factor is not None:
            rois = self.roi_rescale(rois, roi_scale_factor)
        return M.roi_align_levels(feats if isinstance(feats, M.PyramidTaps) else list(feats), rois, lvls, l0.output_size,
                                  [l.spatial_scale for l in self.roi_layers], l0.sampling_ratio, l0.aligned)


BA_ONE_LAUNCH = __import__('os').environ.get('HTD_BA_ONE_LAUNCH', '1') != '0'      # 0: one RoIAlign launch per level (A/B runs)


@ROI_EXTRACTORS.register_module()
class AdptRoIExtractor(BaseRoIExtractor):
    """BA: every RoI is pooled from ALL levels, fused with per-RoI softmax attention over levels, plus the
    outer `edge` ring of the finest level's RoI feature."""

    def __init__(self, aggregation='sum', pre_cfg=None, post_cfg=None, edge=2, **kwargs):
        super().__init__(**kwargs)
        assert aggregation in ['sum', 'concat']
        self.aggregation, self.with_post, self.with_pre, self.edge = aggregation, post_cfg is not None, \
            pre_cfg is not None, edge
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.conv1 = nn.Conv2d(in_channels=256, out_channels=128, kernel_size=1, stride=1)
        self.conv2 = nn.Conv2d(in_channels=128, out_channels=1, kernel_size=1, stride=1)
        self.att = nn.Sequential(self.pool, self.conv1, nn.Tanh(), self.conv2)

    def attention_logits(self, pooled):
        """pooled (L*n, 256) -> (L*n,) : 1x1 conv 256->128, tanh, 1x1 conv 128->1 (:38-46) as two GEMVs."""
        from .. import dense
        h = torch.tanh(dense.linear(pooled, self.conv1.weight.view(128, 256), self.conv1.bias))
        return dense.linear(h, self.conv2.weight.view(1, 128), self.conv2.bias).view(-1)

    def forward(self, feats, rois, roi_scale_factor=None):
        if len(feats) == 1:
            return self.roi_layers[0](feats[0], rois)
        n = rois.size(0)
        out_size = self.roi_layers[0].output_size
        if n == 0:
            return feats[0].new_zeros(0, self.out_channels, *out_size)
        if roi_scale_factor is not None:
            rois = self.roi_rescale(rois, roi_scale_factor)
        L = len(feats)
        l0 = self.roi_layers[0]
        same = all((l.output_size, l.sampling_ratio, l.aligned, l.pool_mode) == (l0.output_size, l0.sampling_ratio, l0.aligned,
                                                                                   l0.pool_mode) for l in self.roi_layers[:L])
        if BA_ONE_LAUNCH and same and rois.is_cuda:
            # every RoI on every level: ONE RoIAlign launch forward, ONE gather launch backward (mmcv_ops._RoIAlignAllLevels)
            lvl_feats = M.roi_align_all_levels(feats, rois, l0.output_size, [l.spatial_scale for l in self.roi_layers[:L]],
                                               l0.sampling_ratio, l0.aligned)
        elif isinstance(feats, M.PyramidTaps):        # chained gradient maps (see mmcv_ops.PyramidTaps)
            lvl_feats = []
            for i in range(L):
                f, feats.levels[i] = self.roi_layers[i](feats.levels[i], rois, chain=True)
                lvl_feats.append(f)
        else:
            lvl_feats = [self.roi_layers[i](feats[i], rois) for i in range(L)]
        if rois.is_cuda and torch.is_grad_enabled() and any(f.requires_grad for f in lvl_feats):
            # a level's features feed the pooling AND the weighted sum below: the sum reads the pooling node's alias, so the two
            # gradients of a level meet inside the pooling's backward (mmcv_ops.GlobalAvgPoolFunction, chain)
            pairs = [M.global_avg_pool(f, chain=True) for f in lvl_feats]
            pooled = torch.cat([p.view(n, -1) for p, _ in pairs], 0)
            lvl_feats = [a for _, a in pairs]
        else:
            pooled = torch.cat([M.global_avg_pool(f).view(n, -1) for f in lvl_feats], 0)
        att = self.attention_logits(pooled).view(L, n)          # n == 1 keeps its axis (reference .squeeze() bug)
        # roi_layers[0](feats[0], rois) of :87 equals lvl_feats[0]: evaluated once
        return M.ba_fuse(att, lvl_feats, self.edge)
