"""The HTD model / train / test configuration as data (configs/htd/htd_resnet50_1x.py:5-168 and its
R101 / R101-DCN siblings htd_resnet101_2x.py, htd_resnet101_dcn_2x_mstrain.py:139-150), produced
programmatically so the package carries no config files of its own.  The reference's config *files*
load unchanged through `htd_amd.Config.fromfile` (same `type=` names and kwargs).
"""
import copy

from .registry import ConfigDict


def _bbox_head(kind, stds, **extra):
    d = dict(type=kind, in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80,
             bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=stds),
             reg_class_agnostic=True,
             loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
             loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
    d.update(extra)
    return d


def _rcnn(thr):
    return dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=thr, neg_iou_thr=thr, min_pos_iou=thr,
                              match_low_quality=False, ignore_iof_thr=-1),
                sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1,
                             add_gt_as_proposals=True),
                pos_weight=-1, debug=False)


def htd_model(depth=50, dcn=False, resnext=False):
    backbone = dict(type='ResNet', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                    norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch')
    if resnext:                                    # htd_resnetx101_dcn_2x_mstrain.py:138-150 (64x4d)
        backbone.update(type='ResNeXt', groups=64, base_width=4)
    if dcn:
        backbone.update(dcn=dict(type='DCN', deform_groups=1, fallback_on_stride=False),
                        stage_with_dcn=(False, True, True, True))
    roi_layer = dict(type='RoIAlign', output_size=7, sampling_ratio=0)
    return dict(
        type='FasterRCNN', pretrained=None, backbone=backbone,
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
        rpn_head=dict(type='RPNHead', in_channels=256, feat_channels=256,
                      anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0],
                                            strides=[4, 8, 16, 32, 64]),
                      bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0],
                                      target_stds=[1.0, 1.0, 1.0, 1.0]),
                      loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                      loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)),
        roi_head=dict(type='HTDRoIHead', num_stages=2, with_global=True, stage_loss_weights=[1, 0.5],
                      bbox_roi_extractor=[
                          dict(type='SingleRoIExtractor', roi_layer=dict(roi_layer), out_channels=256,
                               featmap_strides=[4, 8, 16, 32]),
                          dict(type='AdptRoIExtractor', edge=1, roi_layer=dict(roi_layer), out_channels=256,
                               featmap_strides=[4, 8, 16, 32])],
                      bbox_head=[_bbox_head('Shared2FCBBoxHead', [0.1, 0.1, 0.2, 0.2]),
                                 _bbox_head('HTDBBoxHead', [0.05, 0.05, 0.1, 0.1], relpace=False, edge=1)]))


def htd_train_cfg():
    return dict(
        rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                               match_low_quality=True, ignore_iof_thr=-1),
                 sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1,
                              add_gt_as_proposals=False),
                 allowed_border=0, pos_weight=-1, debug=False),
        rpn_proposal=dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_num=2000, nms_thr=0.7,
                          min_bbox_size=0),
        rcnn=[_rcnn(0.5), _rcnn(0.6)])


def htd_test_cfg(soft_nms=False):
    nms = dict(type='soft_nms', iou_thr=0.5, min_score=0.05) if soft_nms else dict(type='nms', iou_threshold=0.5)
    return dict(rpn=dict(nms_across_levels=False, nms_pre=1000, nms_post=1000, max_num=1000, nms_thr=0.7,
                         min_bbox_size=0),
                rcnn=dict(score_thr=0.05, nms=nms, max_per_img=100))


def htd_config(depth=50, dcn=False, soft_nms=None, resnext=False):
    """-> ConfigDict(model=..., train_cfg=..., test_cfg=..., optimizer=..., lr_config=...)."""
    soft_nms = (depth == 101) if soft_nms is None else soft_nms        # htd_resnet101_2x.py:298
    return ConfigDict(
        model=htd_model(depth, dcn, resnext), train_cfg=htd_train_cfg(), test_cfg=htd_test_cfg(soft_nms),
        optimizer=dict(type='SGD', lr=0.02 if depth == 50 else 0.015, momentum=0.9, weight_decay=0.0001),
        optimizer_config=dict(grad_clip=None),
        lr_config=dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001,
                       step=[8, 11] if depth == 50 else [16, 22]),
        total_epochs=12 if depth == 50 else 24)


def build_htd_detector(depth=50, dcn=False, cfg=None, bf16=False, resnext=False):
    """bf16=True: backbone stages, FPN and the RPN's shared conv run on the bf16 MFMA kernels (fp32 master weights, fp32
    accumulate); the stem, the RoI head and all box / loss arithmetic stay fp32 (BASELINE configs[2] precision map)."""
    from . import detector  # noqa: F401  (registers the components)
    from .registry import build_detector
    cfg = htd_config(depth, dcn, resnext=resnext) if cfg is None else cfg
    # the reference's mixed-precision switch is the config key `fp16 = dict(loss_scale=...)` (mmdet/apis/train.py:97-100,
    # Fp16OptimizerHook); MI355X's 16-bit training type is bf16 (fp32's exponent range: no loss scaling needed), so the key
    # selects this mode and its loss_scale is ignored
    if cfg.get('fp16', None) is not None:
        bf16 = True
    model = build_detector(copy.deepcopy(cfg.model.to_dict()), train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    if bf16:
        import os
        import torch
        if 'HTD_OVERLAP_WGRAD' not in os.environ:
            # weight gradients on a second stream: +2 % on the fp32 step, -6 % on the bf16 one (R101: 33.3 -> 35.4 ms; its
            # hundreds of 10-40 us kernels gain nothing from sharing the chip and pay for the cross-stream waits).  A
            # property of THIS model (runner.Trainer applies it around its steps), not of the process.
            model.overlap_wgrad = False
        model.backbone.compute_dtype = torch.bfloat16
        for head in model.roi_head.bbox_head:          # the 12544->1024->1024 FC stacks of both stages
            head.compute_dtype = torch.bfloat16
            for m in getattr(head, 'convs', []):       # and the 3x3 stack of the regression branch (GroupNorm stays fp32)
                m.compute_dtype = torch.bfloat16
    return model
