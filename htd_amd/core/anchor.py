"""Anchors for the RPN (mmdet/core/anchor/anchor_generator.py:10-340, utils.py:4-46).

Same constructor kwargs and numerics as the reference `AnchorGenerator` (exact fp32 equality is a
tested contract); the per-level grids are cached per (feature-map size, device) instead of being
rebuilt every step (anchor_head.py:159,553 recompute them each call)."""
import numpy as np
import torch

from ..registry import ANCHOR_GENERATORS


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


@ANCHOR_GENERATORS.register_module()
class AnchorGenerator:
    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True, octave_base_scale=None,
                 scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0:
            assert centers is None, f'center cannot be set when center_offset!=0, {centers} is given.'
        if not (0 <= center_offset <= 1):
            raise ValueError(f'center_offset should be in range [0, 1], {center_offset} is given.')
        self.strides = [_pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else base_sizes
        assert len(self.base_sizes) == len(self.strides)
        assert ((octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None)), \
            'scales and octave_base_scale with scales_per_octave cannot be set at the same time'
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            octave = np.array([2**(i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave * octave_base_scale)
        self.octave_base_scale, self.scales_per_octave = octave_base_scale, scales_per_octave
        self.ratios = torch.Tensor(ratios)
        self.scale_major, self.centers, self.center_offset = scale_major, centers, center_offset
        self.base_anchors = [self.gen_single_level_base_anchors(b, self.scales, self.ratios,
                                                                None if centers is None else centers[i])
                             for i, b in enumerate(self.base_sizes)]
        self._cache = {}

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    @property
    def num_levels(self):
        return len(self.strides)

    def gen_single_level_base_anchors(self, base_size, scales, ratios, center=None):
        w = h = base_size
        xc, yc = (self.center_offset * w, self.center_offset * h) if center is None else center
        h_ratios = torch.sqrt(ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
        else:
            ws = (w * scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([xc - 0.5 * ws, yc - 0.5 * hs, xc + 0.5 * ws, yc + 0.5 * hs], dim=-1)

    def single_level_grid_anchors(self, base_anchors, featmap_size, stride=(16, 16), device='cuda'):
        fh, fw = int(featmap_size[0]), int(featmap_size[1])
        sx = torch.arange(0, fw, device=device) * stride[0]
        sy = torch.arange(0, fh, device=device) * stride[1]
        xx = sx.repeat(fh)
        yy = sy.view(-1, 1).repeat(1, fw).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1).type_as(base_anchors)
        return (base_anchors[None, :, :] + shifts[:, None, :]).view(-1, 4)

    def grid_anchors(self, featmap_sizes, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        key = ('a', tuple((int(h), int(w)) for h, w in featmap_sizes), str(device))
        if key not in self._cache:
            self._cache[key] = [self.single_level_grid_anchors(self.base_anchors[i].to(device), featmap_sizes[i],
                                                               self.strides[i], device)
                                for i in range(self.num_levels)]
        return self._cache[key]

    def single_level_valid_flags(self, featmap_size, valid_size, num_base_anchors, device='cuda'):
        fh, fw = featmap_size
        vh, vw = valid_size
        assert vh <= fh and vw <= fw
        vx = torch.zeros(fw, dtype=torch.bool, device=device)
        vy = torch.zeros(fh, dtype=torch.bool, device=device)
        vx[:vw] = 1
        vy[:vh] = 1
        valid = vx.repeat(fh) & vy.view(-1, 1).repeat(1, fw).view(-1)
        return valid[:, None].expand(valid.size(0), num_base_anchors).contiguous().view(-1)

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        key = ('f', tuple((int(h), int(w)) for h, w in featmap_sizes), tuple(int(v) for v in pad_shape[:2]),
               str(device))
        if key not in self._cache:
            flags = []
            h, w = pad_shape[:2]
            for i in range(self.num_levels):
                fh, fw = int(featmap_sizes[i][0]), int(featmap_sizes[i][1])
                vh = min(int(np.ceil(h / self.strides[i][1])), fh)
                vw = min(int(np.ceil(w / self.strides[i][0])), fw)
                flags.append(self.single_level_valid_flags((fh, fw), (vh, vw), self.num_base_anchors[i], device))
            self._cache[key] = flags
        return self._cache[key]

    def __repr__(self):
        return (f'{self.__class__.__name__}(strides={self.strides}, ratios={self.ratios}, scales={self.scales}, '
                f'base_sizes={self.base_sizes}, scale_major={self.scale_major}, num_levels={self.num_levels}, '
                f'centers={self.centers}, center_offset={self.center_offset})')


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags


def images_to_levels(target, num_levels):
    """[per-image (A, ...)] -> [per-level (B, A_l, ...)]."""
    target = torch.stack(target, 0)
    out, start = [], 0
    for n in num_levels:
        out.append(target[:, start:start + n])
        start += n
    return out
