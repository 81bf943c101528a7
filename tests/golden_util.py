"""Deterministic tensors shared by tests/golden/make_golden.py and the tests, so fixtures
hold only small inputs/outputs: every large input or weight is re-created from
numpy RandomState(crc32(name) ^ seed), independent of iteration order."""
import zlib

import numpy as np
import torch


def seeded_array(name, shape, kind='randn', scale=1.0, seed=1234):
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    if kind == 'randn':
        v = rs.randn(*shape) * scale
    elif kind == 'rand':
        v = rs.rand(*shape) * scale
    else:
        raise ValueError(kind)
    return v.astype(np.float32)


def seeded_tensor(name, shape, kind='randn', scale=1.0, seed=1234):
    return torch.from_numpy(seeded_array(name, shape, kind, scale, seed))


def seeded_state_value(name, shape, seed=1234):
    """Non-degenerate value for a state_dict entry, chosen from its key name and shape."""
    shape = tuple(int(s) for s in shape)
    if name.endswith('running_var'):
        return seeded_array(name, shape, 'rand', 0.5, seed) + 0.75
    if name.endswith('running_mean'):
        return seeded_array(name, shape, 'randn', 0.1, seed)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return seeded_array(name, shape, 'randn', float(np.sqrt(1.0 / fan_in)), seed)
    if name.endswith('weight'):
        return 1.0 + seeded_array(name, shape, 'randn', 0.1, seed)
    return seeded_array(name, shape, 'randn', 0.05, seed)


def seeded_state_dict(shapes, prefix='', seed=1234):
    """shapes: mapping key -> shape.  Keys ending in num_batches_tracked are skipped."""
    return {k: torch.from_numpy(seeded_state_value(prefix + k, s, seed))
            for k, s in shapes.items() if not k.endswith('num_batches_tracked')}


def load_seeded_(module, prefix='', seed=1234):
    """In-place: give every parameter/buffer of `module` its seeded value (key = prefix+name)."""
    seen = set()
    live = {t.data_ptr() for t in list(module.parameters()) + list(module.buffers())}
    detached = {}
    with torch.no_grad():
        for k, t in module.state_dict().items():
            # aliases of one tensor (AdptRoIExtractor.conv1 is also att.1): first key wins
            if k.endswith('num_batches_tracked') or t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            v = torch.from_numpy(seeded_state_value(prefix + k, t.shape, seed))
            if t.data_ptr() in live:
                t.copy_(v)
            else:       # state_dict entry is a re-laid-out copy (TileLinear saves its logical 2-D matrix)
                detached[k] = v
    if detached:
        module.load_state_dict(detached, strict=False)
    return module


def digest(t):
    """(sum, abs-sum, a strided sample of <=4096 values) of a tensor, float64."""
    t = t.detach().double().reshape(-1)
    step = max(1, t.numel() // 4096)
    return np.array([t.sum().item(), t.abs().sum().item()]), t[::step][:4096].numpy().copy()


def demo_inputs(B, H, W, rng, num_classes=80):
    """Synthetic COCO-shaped batch; recipe of the reference fixture _demo_mm_inputs
    (tests/test_models/test_forward.py:276-341) with labels in [0, num_classes)."""
    imgs = rng.rand(B, 3, H, W).astype(np.float32)
    gts, labels = [], []
    for _ in range(B):
        k = rng.randint(1, 10)
        cx, cy, bw, bh = rng.rand(k, 4).T
        tl_x = ((cx * W) - (W * bw / 2)).clip(0, W)
        tl_y = ((cy * H) - (H * bh / 2)).clip(0, H)
        br_x = ((cx * W) + (W * bw / 2)).clip(0, W)
        br_y = ((cy * H) + (H * bh / 2)).clip(0, H)
        gts.append(np.vstack([tl_x, tl_y, br_x, br_y]).T.astype(np.float32))
        labels.append(rng.randint(0, num_classes, size=k).astype(np.int64))
    return imgs, gts, labels




def aug_inputs():
    """One image under two test-time augmentations (scale 1.0 unflipped, scale 1.25 flipped); same recipe as
    tests/golden/make_golden.py::aug_inputs, which fed the reference's aug_test for tests/golden/aug_test.npz."""
    rs = np.random.RandomState(5)
    imgs, metas = [], []
    for (H, W), (h, w), sf, flip in [((160, 224), (150, 210), 1.0, False), ((192, 256), (188, 256), 1.25, True)]:
        im = ((rs.rand(1, 3, H, W) - 0.5) * 4).astype(np.float32)
        im[:, :, h:] = 0
        im[:, :, :, w:] = 0
        imgs.append(im)
        metas.append([dict(img_shape=(h, w, 3), pad_shape=(H, W, 3), ori_shape=(150, 210, 3),
                           scale_factor=np.array([sf] * 4, dtype=np.float32), flip=flip,
                           flip_direction='horizontal' if flip else None)])
    return imgs, metas


def match_detections(mine, ref, tol=1e-2):
    """One-to-one matching of (k, 6) [x1, y1, x2, y2, score, class] rows: detections with (nearly) equal scores may
    swap ranks, so rows are matched instead of compared in order."""
    assert mine.shape == ref.shape, (mine.shape, ref.shape)
    used = np.zeros(len(mine), dtype=bool)
    for r in ref:
        d = np.abs(mine[:, :5] - r[:5]).max(1) + 1e3 * (mine[:, 5] != r[5]) + 1e3 * used
        j = int(d.argmin())
        assert d[j] <= tol + 1e-3 * np.abs(r[:4]).max(), (r, mine[j], d[j])
        used[j] = True


def pipeline_samples():
    """Decoded-image stand-ins + annotations of tests/golden/pipeline.npz (same recipe as make_golden.py)."""
    out = []
    for s, (h, w) in enumerate([(120, 160), (200, 150), (97, 333), (64, 64)]):
        rs = np.random.RandomState(s)
        xy = rs.uniform(0, [w - 8, h - 8], (5, 2))
        boxes = np.concatenate([xy, xy + rs.uniform(4, 60, (5, 2))], 1).astype(np.float32)
        img = np.random.RandomState(100 + s).randint(0, 256, (h, w, 3)).astype(np.uint8)
        out.append((img, boxes, rs.randint(0, 80, 5).astype(np.int64)))
    return out


def reference_parameter_order(shapes):
    """`model.parameters()` order of the reference detector = nn.Module registration order, derived from its sources (the
    oracle's `state_shapes` enumerates the same names level by level, which is NOT that order):
      detectors/two_stage.py:26-43       backbone, neck, rpn_head, roi_head
      backbones/resnet.py (Bottleneck)    conv1, bn1, conv2, bn2, conv3, bn3, downsample; stem conv1, bn1; layer1..4
      necks/fpn.py:112-113                the ModuleLists lateral_convs THEN fpn_convs (filled alternately at :134-135,
                                          but a ModuleList lists its own children together)
      dense_heads/rpn_head.py:25-29       rpn_conv, rpn_cls, rpn_reg
      roi_heads/htd_roi_head.py:49-62     bbox_roi_extractor, bbox_head, glbctx_head
      roi_extractors/adaptative_roi_extractor.py:39-46   conv1, conv2 (att.* are the same modules again)
      bbox_heads/bbox_head.py / convfc_bbox_head.py      fc_cls, fc_reg (created by BBoxHead.__init__, re-assigned in
                                          place later), then shared_fcs
      bbox_heads/htd_bbox_head.py:52,73-127              fc_cls, fc_reg, convs, fcs, graph_lvl0..3_cls
      bbox_heads/global_context_head.py:351-373          convs, fc
    BN statistics are buffers, not parameters."""
    names = [k for k in shapes if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]
    first = min(i for i, k in enumerate(names) if k.startswith('neck.'))
    neck = [k for k in names if k.startswith('neck.')]
    assert names[first:first + len(neck)] == neck                   # one contiguous block in the oracle's enumeration
    ordered = [k for k in neck if k.startswith('neck.lateral_convs.')] + [k for k in neck if k.startswith('neck.fpn_convs.')]
    assert len(ordered) == len(neck)
    return names[:first] + ordered + names[first + len(neck):]
