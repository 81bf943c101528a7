"""Registers every component of the HTD path under the reference's registry names."""
from . import losses, resnet, fpn, rpn_head, roi_extractors, bbox_heads, global_context_head, htd_bbox_head  # noqa
from . import htd_roi_head, two_stage  # noqa: F401
from .. import dcn  # noqa: F401  ('DCN' / 'DCNv2' conv layers)
