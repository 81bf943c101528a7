"""GPU: the fused on-device data pipeline (htd_image_batch_pipeline through htd_amd.pipelines.collate) against the CPU
oracle (oracle/pipeline.py) -- bit-exact: the resize is integer arithmetic and the normalisation a float subtract and
one double multiply, both restated operation for operation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def _img(h, w, seed=0):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def _deferred(raw, out_hw, flip=None, pad_hw=None, norm=True, to_rgb=True, pad_val=0.0):
    from htd_amd.pipelines import DeferredImage
    d = DeferredImage(raw)
    d.out_hw, d.flip, d.pad_hw, d.pad_val = out_hw, flip, pad_hw, pad_val
    d.norm = (np.float32(MEAN), np.float32(STD), to_rgb) if norm else None
    return d


def _oracle(raw, out_hw, flip, norm=True, to_rgb=True):
    from oracle import pipeline as P
    out = P.imresize_bilinear_u8(raw, (out_hw[1], out_hw[0]))
    if flip:
        out = P.imflip(out, flip)
    return P.imnormalize(out, MEAN, STD, to_rgb) if norm else out.astype(np.float32)


CASES = [  # (src h,w) -> (dst h,w), flip
    ((48, 64), (80, 107), None), ((60, 45), (107, 80), 'horizontal'), ((33, 97), (33, 97), 'vertical'),
    ((64, 80), (32, 40), 'diagonal'),                       # exactly 2x down: the INTER_AREA shortcut
    ((200, 150), (51, 38), None), ((7, 5), (90, 71), 'horizontal'), ((1, 1), (9, 13), None), ((120, 3), (40, 1), None)]


@pytest.mark.parametrize('pad_mult', [32, 1])
def test_batch_matches_oracle_bit_for_bit(pad_mult):
    from htd_amd.pipelines import DeviceBatchStager
    from oracle import pipeline as P
    imgs, refs = [], []
    for i, (src, dst, flip) in enumerate(CASES):
        raw = _img(*src, seed=i)
        ph = -(-dst[0] // pad_mult) * pad_mult
        pw = -(-dst[1] // pad_mult) * pad_mult
        imgs.append(_deferred(raw, dst, flip, (ph, pw)))
        refs.append(P.impad_to_multiple(_oracle(raw, dst, flip), pad_mult))
    want = P.collate_images(refs)
    got = DeviceBatchStager('cuda:0')(imgs)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_plain_uint8_to_float_and_pad_value():
    """No Resize / Normalize in the pipeline: the kernel is a uint8 -> fp32 copy with padding (identity weights)."""
    from htd_amd.pipelines import DeviceBatchStager
    a, b = _img(19, 23, 1), _img(31, 10, 2)
    got = DeviceBatchStager('cuda:0')([_deferred(a, (19, 23), None, (32, 32), norm=False, pad_val=7.0),
                                       _deferred(b, (31, 10), None, (32, 32), norm=False, pad_val=7.0)]).cpu().numpy()
    np.testing.assert_array_equal(got[0, :, :19, :23], a.transpose(2, 0, 1).astype(np.float32))
    np.testing.assert_array_equal(got[1, :, :31, :10], b.transpose(2, 0, 1).astype(np.float32))
    assert (got[0, :, 19:] == 7).all() and (got[0, :, :, 23:] == 7).all() and (got[1, :, :, 10:] == 7).all()


def test_config_pipeline_end_to_end_and_full_size_properties():
    """configs/_base_/datasets/coco_detection.py train_pipeline on COCO-sized images: against the oracle per image, and
    the size-independent properties -- flipping twice is the identity on the valid region, the padding is exact zeros,
    a constant image stays constant, and re-running the batch is deterministic."""
    from htd_amd.pipelines import build_pipeline, collate
    from oracle import pipeline as P
    cfg = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
           dict(type='Resize', img_scale=(1333, 800), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.5),
           dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True), dict(type='Pad', size_divisor=32),
           dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    pipe = build_pipeline(cfg)
    shapes = [(480, 640), (427, 640), (640, 480), (375, 500)]
    np.random.seed(11)
    samples = []
    for s, (h, w) in enumerate(shapes):
        boxes = np.array([[10, 20, 200, 300], [0, 0, w, h]], np.float32)
        samples.append(pipe(dict(img=_img(h, w, s), img_info=dict(filename=f'{s}.jpg'), img_prefix=None, bbox_fields=[],
                                 ann_info=dict(bboxes=boxes, labels=np.array([3, 7])))))
    data = collate(samples, 'cuda:0')
    assert data['img'].shape == (4, 3, 1088, 1216)
    img = data['img'].cpu().numpy()
    for s, (h, w) in enumerate(shapes):
        meta = data['img_metas'][s]
        ref = P.pipeline_sample(_img(h, w, s), (1333, 800), meta['flip_direction'], MEAN, STD, True, 32)
        nh, nw = ref['img_shape'][:2]
        np.testing.assert_array_equal(img[s, :, :nh, :nw], ref['img'][:nh, :nw].transpose(2, 0, 1))
        assert (img[s, :, nh:] == 0).all() and (img[s, :, :, nw:] == 0).all()
        assert data['gt_bboxes'][s].is_cuda and data['gt_labels'][s].tolist() == [3, 7]
    again = collate(samples, 'cuda:0')['img']
    assert torch.equal(again, data['img'])
    # flip of flip: plan the same images with the opposite flip and mirror the valid region back
    for s in range(4):
        d = samples[s]['img'].copy()
        d.flip = None if d.flip else 'horizontal'
        other = collate([dict(img=d, img_metas={})], 'cuda:0')['img'][0].cpu().numpy()
        nh, nw = samples[s]['img'].out_hw
        np.testing.assert_array_equal(other[:, :nh, :nw][:, :, ::-1], img[s, :, :nh, :nw])
    flat = np.full((500, 375, 3), 128, np.uint8)
    out = collate([pipe(dict(img=flat, img_info=dict(filename='c'), img_prefix=None, bbox_fields=[],
                             ann_info=dict(bboxes=np.zeros((0, 4), np.float32), labels=np.zeros(0, np.int64))))],
                  'cuda:0')['img'][0].cpu().numpy()
    for c, (m, sd) in enumerate(zip(MEAN, STD)):
        vals = np.unique(out[c, :1067, :800])
        assert len(vals) == 1 and abs(vals[0] - (128 - m) / sd) < 1e-6


def test_pipeline_batch_drives_a_train_step():
    """The collated batch is exactly the keyword set of TwoStageDetector.forward_train."""
    from htd_amd.configs import build_htd_detector
    from htd_amd.pipelines import build_pipeline, collate
    torch.manual_seed(0)
    det = build_htd_detector(50).to('cuda:0').train()
    pipe = build_pipeline([dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
                           dict(type='Resize', img_scale=(320, 256), keep_ratio=True),
                           dict(type='RandomFlip', flip_ratio=0.5),
                           dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True), dict(type='Pad', size_divisor=32),
                           dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])])
    np.random.seed(0)
    samples = []
    for s, (h, w) in enumerate([(240, 320), (300, 200)]):
        boxes = np.array([[20, 30, 150, 200], [60, 10, 180, 120]], np.float32)
        samples.append(pipe(dict(img=_img(h, w, s), img_info=dict(filename=f'{s}.jpg'), img_prefix=None, bbox_fields=[],
                                 ann_info=dict(bboxes=boxes, labels=np.array([1, 5])))))
    data = collate(samples, 'cuda:0')
    losses = det.forward_train(**data)
    total = sum(v if torch.is_tensor(v) else sum(v) for k, v in losses.items() if 'loss' in k)
    assert torch.isfinite(total)
    total.backward()


def test_bad_arguments_are_reported():
    import ctypes
    from htd_amd import capi
    x = torch.zeros(64, dtype=torch.uint8, device='cuda:0')
    with pytest.raises(ValueError, match='std'):
        capi.call('htd_image_batch_pipeline', capi.ptr(x), capi.ptr(x), capi.ptr(x), capi.ptr(x), 1, 4, 4, 0., 0., 0.,
                  1., 0., 1., 1, 0., capi.current_stream_ptr())
    with pytest.raises(ValueError, match='null'):
        capi.call('htd_image_batch_pipeline', None, capi.ptr(x), capi.ptr(x), capi.ptr(x), 1, 4, 4, 0., 0., 0., 1., 1.,
                  1., 1, 0., capi.current_stream_ptr())


def test_tta_pipeline_drives_aug_test():
    """MultiScaleFlipAug (test_time_aug.py:8-121) -> collate -> forward_test -> aug_test on one image."""
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.pipelines import build_pipeline, collate
    torch.manual_seed(0)
    cfg = htd_config(50)
    cfg.test_cfg.rpn.update(nms_pre=200, nms_post=100, max_num=100)
    det = build_htd_detector(cfg=cfg).to('cuda:0').eval()
    tta = build_pipeline([dict(type='LoadImageFromFile'),
                          dict(type='MultiScaleFlipAug', img_scale=[(320, 256), (400, 300)], flip=True,
                               transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                                           dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True),
                                           dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']),
                                           dict(type='Collect', keys=['img'])])])
    data = collate([tta(dict(img=_img(240, 320, 4), img_info=dict(filename='t.jpg'), img_prefix=None))], 'cuda:0')
    assert len(data['img']) == 4 and data['img'][0].shape == (1, 3, 256, 320) and data['img'][2].shape == (1, 3, 320, 416)
    assert [m[0]['flip'] for m in data['img_metas']] == [False, True, False, True]
    with torch.no_grad():
        res = det.forward_test(data['img'], data['img_metas'])
    assert len(res) == 1 and len(res[0]) == 80 and all(c.shape[1] == 5 for c in res[0])


def test_device_batch_matches_reference_class_fixture():
    """tests/golden/pipeline.npz (the reference's own transform classes over oracle/pipeline.py's image functions): the
    collated device batch reproduces the stored pixel digest of every sample bit for bit."""
    import os
    from golden_util import digest, pipeline_samples
    from htd_amd.pipelines import build_pipeline, collate
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pipeline.npz'))
    pipe = build_pipeline([dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
                           dict(type='Resize', img_scale=(1333, 800), keep_ratio=True),
                           dict(type='RandomFlip', flip_ratio=0.5), dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True),
                           dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'),
                           dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])])
    np.random.seed(7)
    samples = [pipe(dict(img=img, img_info=dict(filename=f'{i}.jpg'), img_prefix=None, bbox_fields=[],
                         ann_info=dict(bboxes=boxes, labels=labels)))
               for i, (img, boxes, labels) in enumerate(pipeline_samples())]
    batch = collate(samples, 'cuda:0')['img'].cpu()
    for i, s in enumerate(samples):
        ph, pw = s['img_metas']['pad_shape'][:2]
        assert (ph, pw) == tuple(g[f's{i}.pad_shape'][:2])
        hwc = batch[i, :, :ph, :pw].permute(1, 2, 0).contiguous()
        _, sample = digest(hwc)
        np.testing.assert_array_equal(sample, g[f's{i}.img.sample'])
        np.testing.assert_array_equal(s['gt_bboxes'].numpy(), g[f's{i}.gt_bboxes'])
