"""CPU: the data-pipeline oracle (oracle/pipeline.py) against hand-derived cases, and the host planners of
htd_amd.pipelines against the oracle's per-image pipeline (keys, shapes, boxes, random-number consumption).
The resize restates cv2.resize(INTER_LINEAR) on uint8; cv2 is not installed, so parity with it is UNPINNED (see the
oracle header) -- these tests pin the restatement to properties and to values derived by hand from the algorithm."""
import numpy as np
import pytest
import torch

from oracle import pipeline as P

MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def _img(h, w, seed=0):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def test_rescale_size_matches_the_reference_numbers():
    # 1333x800 keep-ratio on the usual COCO shapes (the numbers every mmdet log prints)
    assert P.rescale_size((640, 480), (1333, 800))[0] == (1067, 800)
    assert P.rescale_size((640, 427), (1333, 800))[0] == (1199, 800)
    assert P.rescale_size((500, 375), (1333, 800))[0] == (1067, 800)
    assert P.rescale_size((427, 640), (1333, 800))[0] == (800, 1199)
    assert P.rescale_size((1000, 200), (1333, 800))[0] == (1333, 267)
    with pytest.raises(ValueError):
        P.rescale_size((10, 10), -1.0)


def test_resize_identity_constant_and_hand_values():
    img = _img(13, 17)
    np.testing.assert_array_equal(P.imresize_bilinear_u8(img, (17, 13)), img)        # same size = identity
    flat = np.full((9, 11, 3), 201, np.uint8)
    assert (P.imresize_bilinear_u8(flat, (23, 31)) == 201).all()                      # weights sum to 2048
    assert (P.imresize_bilinear_u8(flat, (5, 4)) == 201).all()
    # 1x2 -> 1x4: fx = (dx+.5)/2-.5 = -.25, .25, .75, 1.25 -> sx,weights = (0;2048,0) (0;1536,512) (0;512,1536) (1;2048,0)
    row = np.array([[[0, 100, 255], [200, 0, 55]]], np.uint8)
    got = P.imresize_bilinear_u8(row, (4, 1))
    want = np.array([[[0, 100, 255], [50, 75, 205], [150, 25, 105], [200, 0, 55]]], np.uint8)
    np.testing.assert_array_equal(got, want)
    # exactly 2x down: the 2x2 mean with round-half-up, (a+b+c+d+2)>>2
    blk = np.array([[[1, 2, 3], [2, 2, 4]], [[0, 3, 5], [0, 3, 6]]], np.uint8)
    np.testing.assert_array_equal(P.imresize_bilinear_u8(blk, (1, 1)), np.array([[[1, 3, 5]]], np.uint8))


@pytest.mark.parametrize('src,dst', [((37, 53), (80, 56)), ((480, 640), (1067, 800)), ((50, 40), (23, 31))])
def test_resize_tracks_ideal_bilinear(src, dst):
    """Within one grey level of the real-valued half-pixel-centre bilinear (torch align_corners=False) when
    up-sampling or mildly down-sampling: the fixed-point tables only round."""
    import torch.nn.functional as F
    img = _img(*src, seed=3)
    got = P.imresize_bilinear_u8(img, dst).astype(np.float64)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
    ref = F.interpolate(t, size=(dst[1], dst[0]), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() <= 1.0 + 1e-9


def test_normalize_arithmetic():
    img = _img(5, 7, seed=1)
    out = P.imnormalize(img, MEAN, STD, to_rgb=True)
    assert out.dtype == np.float32
    m32, s32 = np.float32(MEAN), np.float32(STD)
    y, x, c = 2, 3, 0                                                          # channel 0 of the output is R = input 2
    want = np.float32(np.float64(np.float32(img[y, x, 2]) - m32[c]) * (1.0 / np.float64(s32[c])))
    assert out[y, x, c] == want
    np.testing.assert_allclose(out, (img[..., ::-1].astype(np.float64) - np.float64(MEAN)) / np.float64(STD),
                               rtol=0, atol=2e-6)


def test_pad_flip_collate_and_boxes():
    a, b = _img(30, 45).astype(np.float32), _img(61, 20, 1).astype(np.float32)
    pa, pb = P.impad_to_multiple(a, 32), P.impad_to_multiple(b, 32)
    assert pa.shape == (32, 64, 3) and pb.shape == (64, 32, 3)
    assert (pa[30:] == 0).all() and (pa[:, 45:] == 0).all()
    batch = P.collate_images([pa, pb])
    assert batch.shape == (2, 3, 64, 64)
    np.testing.assert_array_equal(batch[0, :, :30, :45], a.transpose(2, 0, 1))
    assert (batch[0, :, 32:] == 0).all() and (batch[1, :, :, 32:] == 0).all()
    img = _img(4, 6)
    np.testing.assert_array_equal(P.imflip(P.imflip(img, 'diagonal'), 'diagonal'), img)
    np.testing.assert_array_equal(P.imflip(img, 'diagonal'), P.imflip(P.imflip(img, 'horizontal'), 'vertical'))
    boxes = np.array([[10., 20., 30., 60.]], np.float32)
    np.testing.assert_array_equal(P.bbox_flip(boxes, (100, 200, 3), 'horizontal'), [[170., 20., 190., 60.]])
    np.testing.assert_array_equal(P.bbox_flip(boxes, (100, 200, 3), 'vertical'), [[10., 40., 30., 80.]])
    sf = np.array([2., 1.5, 2., 1.5], np.float32)
    np.testing.assert_array_equal(P.resize_bboxes(boxes, sf, (80, 50, 3)), [[20., 30., 50., 80.]])


# ------------------------------------------------------------------------------ host planners vs the oracle pipeline
def _train_cfg(scale=(1333, 800), **kw):
    return [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
            dict(type='Resize', img_scale=scale, keep_ratio=True, **kw), dict(type='RandomFlip', flip_ratio=0.5),
            dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True), dict(type='Pad', size_divisor=32),
            dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]


def _sample(h, w, seed):
    rs = np.random.RandomState(seed)
    xy = rs.uniform(0, [w - 8, h - 8], (5, 2))
    boxes = np.concatenate([xy, xy + rs.uniform(4, 60, (5, 2))], 1).astype(np.float32)
    return dict(img=_img(h, w, seed), img_info=dict(filename=f'{seed}.jpg'), img_prefix=None,
                ann_info=dict(bboxes=boxes, labels=rs.randint(0, 80, 5).astype(np.int64)), bbox_fields=[])


def test_planners_reproduce_the_reference_result_dict():
    from htd_amd.pipelines import build_pipeline
    pipe = build_pipeline(_train_cfg())
    np.random.seed(7)
    outs = [pipe(_sample(h, w, s)) for s, (h, w) in enumerate([(120, 160), (200, 150), (97, 333), (64, 64)])]
    np.random.seed(7)                       # the reference draws one np.random.choice per image in RandomFlip
    for s, ((h, w), out) in enumerate(zip([(120, 160), (200, 150), (97, 333), (64, 64)], outs)):
        raw = _sample(h, w, s)
        direction = np.random.choice(['horizontal', None], p=[0.5, 0.5])
        ref = P.pipeline_sample(raw['img'], (1333, 800), direction, MEAN, STD, True, 32, raw['ann_info']['bboxes'])
        meta = out['img_metas']
        assert meta['flip'] == ref['flip'] and meta['flip_direction'] == ref['flip_direction']
        assert tuple(meta['img_shape']) == tuple(ref['img_shape'])
        assert tuple(meta['pad_shape']) == tuple(ref['pad_shape'])
        assert tuple(meta['ori_shape']) == (h, w, 3)
        np.testing.assert_array_equal(meta['scale_factor'], ref['scale_factor'])
        assert meta['scale_factor'].dtype == np.float32
        np.testing.assert_array_equal(out['gt_bboxes'].numpy(), ref['gt_bboxes'])
        assert out['gt_bboxes'].dtype == torch.float32 and out['gt_labels'].dtype == torch.int64
        assert out['img'].shape == ref['img'].shape
        assert set(meta) == {'filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip',
                             'flip_direction', 'img_norm_cfg'}


def test_multiscale_range_and_test_time_aug_planning():
    from htd_amd.pipelines import build_pipeline
    # htd_resnet101_dcn_2x_mstrain.py:9-11: img_scale=[(1600,400),(1600,1400)], multiscale_mode='range'
    pipe = build_pipeline(_train_cfg(scale=[(1600, 400), (1600, 1400)], multiscale_mode='range'))
    np.random.seed(3)
    out = pipe(_sample(240, 320, 0))
    np.random.seed(3)
    long_edge = np.random.randint(1600, 1601)
    short_edge = np.random.randint(400, 1401)
    (nw, nh), _ = P.rescale_size((320, 240), (long_edge, short_edge))
    assert tuple(out['img_metas']['img_shape']) == (nh, nw, 3)
    tta = build_pipeline([dict(type='LoadImageFromFile'),
                          dict(type='MultiScaleFlipAug', img_scale=[(1333, 800), (1000, 600)], flip=True,
                               transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                                           dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True),
                                           dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']),
                                           dict(type='Collect', keys=['img'])])])
    res = tta(dict(img=_img(100, 150), img_info=dict(filename='a.jpg'), img_prefix=None))
    assert len(res['img']) == 4 and [m['flip'] for m in res['img_metas']] == [False, True, False, True]
    assert res['img'][0].shape == (800, 1216, 3) and res['img'][2].shape == (608, 928, 3)
    assert res['img'][1].flip == 'horizontal' and res['img'][0].flip is None
    assert res['img'][0].raw is res['img'][3].raw                       # the pixels are shared, never copied per aug


def test_out_of_order_pipelines_are_refused():
    from htd_amd.pipelines import build_pipeline
    bad = build_pipeline([dict(type='LoadImageFromFile'), dict(type='Normalize', mean=MEAN, std=STD),
                          dict(type='Resize', img_scale=(64, 64))])
    with pytest.raises(ValueError, match='Resize'):
        bad(dict(img=_img(10, 10), img_info=dict(filename='x'), img_prefix=None))
    with pytest.raises(TypeError):
        build_pipeline([dict(type='LoadImageFromFile')])(dict(img=np.zeros((4, 4, 3), np.float32), img_info={}))


def _ref_cfg(scale=(1333, 800), **kw):
    return [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
            dict(type='Resize', img_scale=scale, keep_ratio=True, **kw), dict(type='RandomFlip', flip_ratio=0.5),
            dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True), dict(type='Pad', size_divisor=32)]


def _ref_input(img, boxes, labels, i):
    return dict(img=img, img_info=dict(filename=f'{i}.jpg'), img_prefix=None, bbox_fields=[],
                ann_info=dict(bboxes=boxes, labels=labels))


def test_planners_match_the_reference_transform_classes(golden):
    """tests/golden/pipeline.npz was produced by the reference's own Resize / RandomFlip / Normalize / Pad /
    MultiScaleFlipAug classes (make_golden.py pipeline; their mmcv image calls served by oracle/pipeline.py): same
    random draws, shapes, scale factors, flips and boxes from the planners here, and the oracle's per-image pipeline
    reproduces the stored pixel digests."""
    from golden_util import digest, pipeline_samples
    from htd_amd.pipelines import build_pipeline
    g = golden('pipeline')
    pipe = build_pipeline(_ref_cfg())
    np.random.seed(7)
    for i, (img, boxes, labels) in enumerate(pipeline_samples()):
        r = pipe(_ref_input(img, boxes, labels, i))
        assert tuple(r['img_shape']) == tuple(g[f's{i}.img_shape'])
        assert tuple(r['pad_shape']) == tuple(g[f's{i}.pad_shape'])
        np.testing.assert_array_equal(r['scale_factor'], g[f's{i}.scale_factor'])
        assert int(bool(r['flip'])) == int(g[f's{i}.flip'])
        np.testing.assert_array_equal(r['gt_bboxes'], g[f's{i}.gt_bboxes'])
        ref = P.pipeline_sample(img, (1333, 800), r['flip_direction'] if r['flip'] else None, MEAN, STD, True, 32)
        sums, sample = digest(torch.from_numpy(np.ascontiguousarray(ref['img'])))
        np.testing.assert_array_equal(sample, g[f's{i}.img.sample'])
    ms = build_pipeline(_ref_cfg(scale=[(1600, 400), (1600, 1400)], multiscale_mode='range')[:4])
    np.random.seed(3)
    for i, (img, boxes, labels) in enumerate(pipeline_samples()[:2]):
        r = ms(_ref_input(img, boxes, labels, i))
        assert tuple(r['img_shape']) == tuple(g[f'm{i}.img_shape'])
        assert int(bool(r['flip'])) == int(g[f'm{i}.flip'])
        np.testing.assert_array_equal(r['gt_bboxes'], g[f'm{i}.gt_bboxes'])
    tta = build_pipeline([dict(type='LoadImageFromFile'),
                          dict(type='MultiScaleFlipAug', img_scale=[(1333, 800), (1000, 600)], flip=True,
                               transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                                           dict(type='Pad', size_divisor=32)])])
    r = tta(dict(img=pipeline_samples()[0][0], img_info=dict(filename='a.jpg'), img_prefix=None))
    np.testing.assert_array_equal(np.array(r['img_shape']), g['tta.img_shapes'])
    np.testing.assert_array_equal(np.array(r['pad_shape']), g['tta.pad_shapes'])
    assert [int(bool(f)) for f in r['flip']] == g['tta.flips'].tolist()
