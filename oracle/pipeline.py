"""CPU ORACLE (test infrastructure, not product code) of the image side of the reference data pipeline:
Resize(keep_ratio) -> RandomFlip -> Normalize -> Pad(size_divisor) -> collate, i.e. SURVEY 8(f) row 2, the step
immediately before the hot path.  numpy restatement of

  * mmdet/datasets/pipelines/transforms.py:202-231 (Resize._resize_img), :233-241 (_resize_bboxes),
    :381-413 (RandomFlip.bbox_flip), :440-444 (image flip), :496-505 (Pad._pad_img), :563-575 (Normalize);
  * configs/_base_/datasets/coco_detection.py:3-14 (the order and the constants);
  * the mmcv (>= 1.1.5, requirements of the reference; NOT vendored under /root/reference) functions those call:
    mmcv.imrescale / rescale_size, mmcv.imflip, mmcv.imnormalize, mmcv.impad_to_multiple, mmcv.parallel.collate;
  * OpenCV's cv2.resize(INTER_LINEAR) on 8-bit images, which mmcv.imresize calls (cv2 is not installed here):
    the published fixed-point algorithm of modules/imgproc/src/resize.cpp -- coefficient tables in float scaled to
    INTER_RESIZE_COEF_BITS = 11 bits, horizontal pass to int32, vertical pass
    ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2, and the exact-2x-downscale shortcut to INTER_AREA.

PARITY UNPINNED for the PIXELS of the cv2-backed steps (resize, normalize arithmetic): neither mmcv nor OpenCV is
importable in this image and the reference's tests hold no golden pixels for them, so this restatement is checked only
against hand-derived cases (tests/test_pipeline_oracle.py).  Everything else -- scale selection, output sizes, scale
factors, flips, box arithmetic, padding, test-time-augmentation planning -- is pinned by tests/golden/pipeline.npz,
produced by the reference's own transform classes running over this module's image functions (make_golden.py pipeline).
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def rescale_size(old_size, scale):
    """mmcv.image.geometric.rescale_size: (w, h), scale (float | (long, short)) -> ((new_w, new_h), factor)."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        factor = scale
    else:
        max_long, max_short = max(scale), min(scale)
        factor = min(max_long / max(h, w), max_short / min(h, w))
    return (int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)), factor


def _sat_short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)        # cvRound = round-half-even


def linear_coeffs(dst, src, clamp_last):
    """Per destination index: source index and the two 11-bit integer weights of cv2's INTER_LINEAR tables.
    clamp_last=True is the x rule (fx reset to 0 at both borders), False the y rule (rows clipped when read)."""
    scale = 1.0 / (float(dst) / float(src))                            # double, as resize() computes it
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f)
    f = (f - s).astype(np.float32)
    s = s.astype(np.int64)
    if clamp_last:
        lo, hi = s < 0, s >= src - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, src - 1, s))
    w0 = _sat_short((np.float32(1) - f) * np.float32(COEF_SCALE))
    w1 = _sat_short(f * np.float32(COEF_SCALE))
    return s, w0, w1


def imresize_bilinear_u8(img, size):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HWC images."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    dw, dh = size
    sh, sw = img.shape[:2]
    if sw == 2 * dw and sh == 2 * dh:                                  # INTER_LINEAR == fast INTER_AREA at exactly 2x
        v = img.astype(np.int32)
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = linear_coeffs(dw, sw, True)
    sy, b0, b1 = linear_coeffs(dh, sh, False)
    src = img.astype(np.int32)
    sx1 = np.minimum(sx + 1, sw - 1)                                   # weight is 0 wherever this clamps
    rows0 = np.clip(sy, 0, sh - 1)
    rows1 = np.clip(sy + 1, 0, sh - 1)
    hor = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]          # (sh, dw, c) int32
    s0, s1 = hor[rows0], hor[rows1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)                                         # uchar(...) cast; the value is in [0, 255]


def imrescale(img, scale):
    """mmcv.imrescale(img, scale, return_scale=True) with the bilinear cv2 backend."""
    h, w = img.shape[:2]
    new_size, factor = rescale_size((w, h), scale)
    return imresize_bilinear_u8(img, new_size), factor


def imflip(img, direction='horizontal'):
    if direction == 'horizontal':
        return np.flip(img, axis=1)
    if direction == 'vertical':
        return np.flip(img, axis=0)
    if direction == 'diagonal':
        return np.flip(img, axis=(0, 1))
    raise ValueError(f"Invalid flipping direction '{direction}'")


def imnormalize(img, mean, std, to_rgb=True):
    """mmcv.imnormalize: float32 copy, optional BGR->RGB, cv2.subtract(img, mean) in float32 (a non-integer scalar
    against a 32F array is applied in 32F), cv2.multiply(img, 1/float64(std)) in float64 rounded to float32."""
    mean = np.asarray(mean, dtype=np.float32)
    std = np.asarray(std, dtype=np.float32)
    out = img.astype(np.float32)
    if to_rgb:
        out = out[..., ::-1]
    out = out - mean.reshape(1, 1, -1)
    stdinv = 1.0 / std.astype(np.float64)
    return (out.astype(np.float64) * stdinv.reshape(1, 1, -1)).astype(np.float32)


def impad_to_multiple(img, divisor, pad_val=0):
    h, w = img.shape[:2]
    ph = int(np.ceil(h / divisor)) * divisor
    pw = int(np.ceil(w / divisor)) * divisor
    out = np.full((ph, pw) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:h, :w] = img
    return out


def resize_bboxes(bboxes, scale_factor, img_shape):
    """transforms.py:233-241 with bbox_clip_border=True."""
    b = bboxes * scale_factor
    b[:, 0::2] = np.clip(b[:, 0::2], 0, img_shape[1])
    b[:, 1::2] = np.clip(b[:, 1::2], 0, img_shape[0])
    return b


def bbox_flip(bboxes, img_shape, direction):
    """transforms.py:381-413."""
    out = bboxes.copy()
    h, w = img_shape[:2]
    if direction in ('horizontal', 'diagonal'):
        out[..., 0::4] = w - bboxes[..., 2::4]
        out[..., 2::4] = w - bboxes[..., 0::4]
    if direction in ('vertical', 'diagonal'):
        out[..., 1::4] = h - bboxes[..., 3::4]
        out[..., 3::4] = h - bboxes[..., 1::4]
    return out


def pipeline_sample(img, scale, flip_direction, mean, std, to_rgb=True, size_divisor=32, gt_bboxes=None):
    """One image through Resize(keep_ratio) -> RandomFlip(decided) -> Normalize -> Pad; returns the reference's
    result-dict fields (img is HWC float32 here; DefaultFormatBundle's transpose happens in collate_images)."""
    h, w = img.shape[:2]
    out, _ = imrescale(img, scale)
    nh, nw = out.shape[:2]
    scale_factor = np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32)
    res = dict(ori_shape=img.shape, img_shape=out.shape, scale_factor=scale_factor, flip=flip_direction is not None,
               flip_direction=flip_direction)
    if gt_bboxes is not None:
        gt_bboxes = resize_bboxes(np.asarray(gt_bboxes, dtype=np.float32), scale_factor, out.shape)
    if flip_direction is not None:
        out = imflip(out, flip_direction)
        if gt_bboxes is not None:
            gt_bboxes = bbox_flip(gt_bboxes, res['img_shape'], flip_direction)
    out = imnormalize(out, mean, std, to_rgb)
    out = impad_to_multiple(out, size_divisor) if size_divisor else out
    res.update(img=out, pad_shape=out.shape, gt_bboxes=gt_bboxes)
    return res


def collate_images(imgs, pad_val=0.0):
    """DefaultFormatBundle (HWC -> CHW) + mmcv.parallel.collate of stacked DataContainers with pad_dims=2:
    every image is padded at the bottom / right to the largest H, W of the batch."""
    H = max(i.shape[0] for i in imgs)
    W = max(i.shape[1] for i in imgs)
    out = np.full((len(imgs), imgs[0].shape[2], H, W), pad_val, dtype=np.float32)
    for b, i in enumerate(imgs):
        out[b, :, :i.shape[0], :i.shape[1]] = i.transpose(2, 0, 1)
    return out
