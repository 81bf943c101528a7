from .anchor import AnchorGenerator, anchor_inside_flags, images_to_levels  # noqa: F401
from .bbox import (AssignResult, BboxOverlaps2D, DeltaXYWHBBoxCoder, MaxIoUAssigner, RandomSampler,  # noqa: F401
                   SamplingResult, bbox2delta, bbox2result, bbox2roi, bbox_overlaps, delta2bbox, set_randperm)
from .post_processing import merge_aug_bboxes, merge_aug_proposals, merge_aug_scores, multiclass_nms  # noqa: F401
from .misc import multi_apply, unmap  # noqa: F401
