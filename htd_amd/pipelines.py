"""Data pipeline of the reference configs with the pixel work moved onto the MI355X.

The reference runs Resize -> RandomFlip -> Normalize -> Pad -> DefaultFormatBundle -> Collect per image on CPU workers
(configs/_base_/datasets/coco_detection.py:5-28; mmdet/datasets/pipelines/{transforms,formating,loading,
test_time_aug,compose}.py) and mmcv.parallel.collate pads the float32 images of a batch to a common shape: 12 B per
padded pixel cross PCIe.  Here the transforms keep their names, arguments, random-number consumption and every
result-dict key, but they are *planners*: they decide geometry (output size, flip, normalisation constants, padding) and
transform the boxes on the host (a few floats), while the decoded uint8 pixels stay untouched in `results['img']`
(a `DeferredImage`).  `collate()` then uploads the raw bytes of the whole batch (3 B per SOURCE pixel, one pinned copy)
and one launch of htd_image_batch_pipeline writes the collated, normalised, zero-padded NHWC fp32 batch.

The fused kernel fixes the order resize -> flip -> normalize -> pad (the order of every reference config); a pipeline
that asks for another order raises instead of silently computing something else.
"""
import os.path as osp
import warnings

import numpy as np
import torch

from . import capi
from .registry import Registry, build_from_cfg

PIPELINES = Registry('pipeline')

_FLIP_BITS = {None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}


class DeferredImage:
    """The decoded uint8 HWC image plus the geometry the transforms have decided so far."""

    def __init__(self, raw):
        raw = np.ascontiguousarray(raw)
        if raw.dtype != np.uint8 or raw.ndim != 3 or raw.shape[2] != 3:
            raise TypeError(f'the device pipeline takes HxWx3 uint8 images, got {raw.dtype} {raw.shape}')
        self.raw = raw
        self.out_hw = raw.shape[:2]            # after Resize
        self.flip = None                       # 'horizontal' | 'vertical' | 'diagonal' | None
        self.norm = None                       # (mean, std, to_rgb)
        self.pad_hw = None                     # after Pad
        self.pad_val = 0.0
        self.stage = 0                         # 0 loaded, 1 resized, 2 flipped, 3 normalised, 4 padded

    def advance(self, stage, what):
        if self.stage >= stage:
            raise ValueError(f'{what} after stage {self.stage}: the fused device pipeline runs Resize -> RandomFlip '
                             '-> Normalize -> Pad, each at most once and in this order')
        self.stage = stage

    @property
    def shape(self):                           # what results['img'].shape reads in the reference at this point
        h, w = self.pad_hw or self.out_hw
        return (h, w, 3)

    def copy(self):
        other = DeferredImage.__new__(DeferredImage)
        other.__dict__.update(self.__dict__)
        return other


def rescale_size(old_size, scale):
    """mmcv.rescale_size((w, h), scale) -> ((new_w, new_h), factor)."""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        factor = scale
    elif isinstance(scale, tuple):
        factor = min(max(scale) / max(h, w), min(scale) / min(h, w))
    else:
        raise TypeError(f'Scale must be a number or tuple of int, but got {type(scale)}')
    return (int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)), factor


@PIPELINES.register_module()
class LoadImageFromFile:
    """loading.py:12-68.  Takes an in-memory array (`results['img']`, BGR like mmcv.imread) or decodes the file with
    PIL; JPEG decoders differ by +-1 between libraries, so decoded pixels are outside the parity claim."""

    def __init__(self, to_float32=False, color_type='color', file_client_args=None):
        if to_float32:
            raise NotImplementedError('the device pipeline consumes uint8 pixels (to_float32=False)')
        self.color_type = color_type

    def __call__(self, results):
        info = results.get('img_info', {})
        name = info.get('filename')
        filename = osp.join(results['img_prefix'], name) if results.get('img_prefix') is not None and name else name
        img = results.get('img')
        if img is None:
            from PIL import Image
            with Image.open(filename) as im:
                img = np.asarray(im.convert('RGB'))[:, :, ::-1]
        results['filename'] = filename
        results['ori_filename'] = name
        results['img'] = DeferredImage(img)
        results['img_shape'] = results['img'].raw.shape
        results['ori_shape'] = results['img'].raw.shape
        results['img_fields'] = ['img']
        return results


@PIPELINES.register_module()
class LoadAnnotations:
    """loading.py:196-324, boxes and labels (HTD trains on boxes only)."""

    def __init__(self, with_bbox=True, with_label=True, with_mask=False, with_seg=False, poly2mask=True,
                 file_client_args=None):
        if with_mask or with_seg:
            raise NotImplementedError('masks / semantic maps are outside the HTD path')
        self.with_bbox, self.with_label = with_bbox, with_label

    def __call__(self, results):
        ann = results['ann_info']
        results.setdefault('bbox_fields', [])
        if self.with_bbox:
            results['gt_bboxes'] = ann['bboxes'].copy()
            ignore = ann.get('bboxes_ignore', None)
            if ignore is not None:
                results['gt_bboxes_ignore'] = ignore.copy()
                results['bbox_fields'].append('gt_bboxes_ignore')
            results['bbox_fields'].append('gt_bboxes')
        if self.with_label:
            results['gt_labels'] = ann['labels'].copy()
        return results


@PIPELINES.register_module()
class Resize:
    """transforms.py:25-315: scale selection (same np.random calls), output size, scale_factor, box scaling."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True,
                 bbox_clip_border=True, backend='cv2', override=False):
        if img_scale is None:
            self.img_scale = None
        else:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            assert all(isinstance(s, tuple) for s in self.img_scale)
        if ratio_range is not None:
            assert len(self.img_scale) == 1
        else:
            assert multiscale_mode in ['value', 'range']
        if backend != 'cv2':
            raise NotImplementedError("only the cv2 bilinear resize of the reference configs is implemented")
        self.multiscale_mode, self.ratio_range, self.keep_ratio = multiscale_mode, ratio_range, keep_ratio
        self.override, self.bbox_clip_border = override, bbox_clip_border

    @staticmethod
    def random_select(img_scales):
        idx = np.random.randint(len(img_scales))
        return img_scales[idx], idx

    @staticmethod
    def random_sample(img_scales):
        assert len(img_scales) == 2
        longs, shorts = [max(s) for s in img_scales], [min(s) for s in img_scales]
        long_edge = np.random.randint(min(longs), max(longs) + 1)
        short_edge = np.random.randint(min(shorts), max(shorts) + 1)
        return (long_edge, short_edge), None

    @staticmethod
    def random_sample_ratio(img_scale, ratio_range):
        lo, hi = ratio_range
        assert lo <= hi
        ratio = np.random.random_sample() * (hi - lo) + lo
        return (int(img_scale[0] * ratio), int(img_scale[1] * ratio)), None

    def _random_scale(self, results):
        if self.ratio_range is not None:
            scale, idx = self.random_sample_ratio(self.img_scale[0], self.ratio_range)
        elif len(self.img_scale) == 1:
            scale, idx = self.img_scale[0], 0
        elif self.multiscale_mode == 'range':
            scale, idx = self.random_sample(self.img_scale)
        else:
            scale, idx = self.random_select(self.img_scale)
        results['scale'], results['scale_idx'] = scale, idx

    def __call__(self, results):
        if 'scale' not in results:
            if 'scale_factor' in results:
                f = results['scale_factor']
                assert isinstance(f, float)
                results['scale'] = tuple([int(x * f) for x in results['img'].shape[:2]][::-1])
            else:
                self._random_scale(results)
        elif not self.override:
            assert 'scale_factor' not in results, 'scale and scale_factor cannot be both set.'
        else:
            results.pop('scale')
            results.pop('scale_factor', None)
            self._random_scale(results)
        img = results['img'] = results['img'].copy()
        img.advance(1, 'Resize')
        h, w = img.raw.shape[:2]
        if self.keep_ratio:
            (new_w, new_h), _ = rescale_size((w, h), results['scale'])
        else:
            new_w, new_h = results['scale']
        img.out_hw = (new_h, new_w)
        w_scale, h_scale = new_w / w, new_h / h
        results['scale_factor'] = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
        results['img_shape'] = results['pad_shape'] = (new_h, new_w, 3)
        results['keep_ratio'] = self.keep_ratio
        for key in results.get('bbox_fields', []):
            boxes = results[key] * results['scale_factor']
            if self.bbox_clip_border:
                boxes[:, 0::2] = np.clip(boxes[:, 0::2], 0, new_w)
                boxes[:, 1::2] = np.clip(boxes[:, 1::2], 0, new_h)
            results[key] = boxes
        return results


@PIPELINES.register_module()
class RandomFlip:
    """transforms.py:318-472."""

    def __init__(self, flip_ratio=None, direction='horizontal'):
        if isinstance(flip_ratio, list):
            assert 0 <= sum(flip_ratio) <= 1
        elif flip_ratio is not None:
            assert 0 <= flip_ratio <= 1
        valid = ['horizontal', 'vertical', 'diagonal']
        assert direction in valid if isinstance(direction, str) else set(direction).issubset(valid)
        if isinstance(flip_ratio, list):
            assert len(flip_ratio) == len(direction)
        self.flip_ratio, self.direction = flip_ratio, direction

    @staticmethod
    def bbox_flip(bboxes, img_shape, direction):
        assert bboxes.shape[-1] % 4 == 0
        if direction not in ('horizontal', 'vertical', 'diagonal'):
            raise ValueError(f"Invalid flipping direction '{direction}'")
        out = bboxes.copy()
        h, w = img_shape[:2]
        if direction != 'vertical':
            out[..., 0::4] = w - bboxes[..., 2::4]
            out[..., 2::4] = w - bboxes[..., 0::4]
        if direction != 'horizontal':
            out[..., 1::4] = h - bboxes[..., 3::4]
            out[..., 3::4] = h - bboxes[..., 1::4]
        return out

    def __call__(self, results):
        cur_dir = None
        if 'flip' not in results:
            dirs = (self.direction if isinstance(self.direction, list) else [self.direction]) + [None]
            if isinstance(self.flip_ratio, list):
                probs = self.flip_ratio + [1 - sum(self.flip_ratio)]
            else:
                probs = [self.flip_ratio / (len(dirs) - 1)] * (len(dirs) - 1) + [1 - self.flip_ratio]
            cur_dir = np.random.choice(dirs, p=probs)
            results['flip'] = cur_dir is not None
        if 'flip_direction' not in results:
            results['flip_direction'] = cur_dir
        img = results['img'] = results['img'].copy()
        img.advance(2, 'RandomFlip')
        if results['flip']:
            img.flip = results['flip_direction']
            if img.flip not in _FLIP_BITS:
                raise ValueError(f"Invalid flipping direction '{img.flip}'")
            for key in results.get('bbox_fields', []):
                results[key] = self.bbox_flip(results[key], results['img_shape'], results['flip_direction'])
        return results


@PIPELINES.register_module()
class Normalize:
    """transforms.py:546-584."""

    def __init__(self, mean, std, to_rgb=True):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb
        assert self.mean.shape == (3,) and self.std.shape == (3,)

    def __call__(self, results):
        img = results['img'] = results['img'].copy()
        img.advance(3, 'Normalize')
        img.norm = (self.mean, self.std, bool(self.to_rgb))
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results


@PIPELINES.register_module()
class Pad:
    """transforms.py:475-543."""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert (size is None) != (size_divisor is None)
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val

    def __call__(self, results):
        img = results['img'] = results['img'].copy()
        img.advance(4, 'Pad')
        h, w = img.out_hw
        if self.size is not None:
            ph, pw = self.size
            assert ph >= h and pw >= w
        else:
            d = self.size_divisor
            ph, pw = int(np.ceil(h / d)) * d, int(np.ceil(w / d)) * d
        img.pad_hw, img.pad_val = (ph, pw), float(self.pad_val)
        results['pad_shape'] = (ph, pw, 3)
        results['pad_fixed_size'], results['pad_size_divisor'] = self.size, self.size_divisor
        return results


@PIPELINES.register_module()
class DefaultFormatBundle:
    """formating.py:165-230: boxes / labels become tensors; the image stays deferred until collate()."""

    def __call__(self, results):
        results.setdefault('pad_shape', results['img'].shape)
        results.setdefault('scale_factor', 1.0)
        results.setdefault('img_norm_cfg', dict(mean=np.zeros(3, dtype=np.float32), std=np.ones(3, dtype=np.float32),
                                                to_rgb=False))
        for key in ('proposals', 'gt_bboxes', 'gt_bboxes_ignore', 'gt_labels'):
            if key in results:
                results[key] = torch.from_numpy(np.ascontiguousarray(results[key]))
        return results


@PIPELINES.register_module()
class ImageToTensor:
    """formating.py:52-82: HWC -> CHW happens in the kernel's output layout; nothing to do per sample."""

    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        return results


@PIPELINES.register_module()
class Collect:
    """formating.py:233-322."""

    def __init__(self, keys, meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape',
                                        'scale_factor', 'flip', 'flip_direction', 'img_norm_cfg')):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {'img_metas': {k: results[k] for k in self.meta_keys}}
        for key in self.keys:
            data[key] = results[key]
        return data


class Compose:
    """compose.py:8-51."""

    def __init__(self, transforms):
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]
        for t in self.transforms:
            if not callable(t):
                raise TypeError('transform must be callable or a dict')

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


@PIPELINES.register_module()
class MultiScaleFlipAug:
    """test_time_aug.py:8-121."""

    def __init__(self, transforms, img_scale=None, scale_factor=None, flip=False, flip_direction='horizontal'):
        self.transforms = Compose(transforms)
        assert (img_scale is None) ^ (scale_factor is None), 'Must have but only one variable can be setted'
        if img_scale is not None:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            self.scale_key = 'scale'
        else:
            self.img_scale = scale_factor if isinstance(scale_factor, list) else [scale_factor]
            self.scale_key = 'scale_factor'
        self.flip = flip
        self.flip_direction = flip_direction if isinstance(flip_direction, list) else [flip_direction]
        if not self.flip and self.flip_direction != ['horizontal']:
            warnings.warn('flip_direction has no effect when flip is set to False')

    def __call__(self, results):
        flip_args = [(False, None)] + ([(True, d) for d in self.flip_direction] if self.flip else [])
        aug = []
        for scale in self.img_scale:
            for flip, direction in flip_args:
                r = results.copy()
                r[self.scale_key] = scale
                r['flip'], r['flip_direction'] = flip, direction
                aug.append(self.transforms(r))
        return {key: [d[key] for d in aug] for key in aug[0]}


def build_pipeline(cfgs):
    """The `train_pipeline` / `test_pipeline` list of a reference config -> Compose."""
    return Compose(cfgs)


# ------------------------------------------------------------------------------------------- device side
def _plan(images):
    """(packed uint8 + offsets + meta) of a list of DeferredImage, and the collated shape."""
    norm = images[0].norm or (np.zeros(3, np.float32), np.ones(3, np.float32), False)
    pad_val = images[0].pad_val
    metas, offs, total = [], [], 0
    for im in images:
        other = im.norm or norm
        if not (np.array_equal(other[0], norm[0]) and np.array_equal(other[1], norm[1]) and other[2] == norm[2]
                and im.pad_val == pad_val):
            raise ValueError('one batch, one Normalize / Pad configuration')
        sh, sw = im.raw.shape[:2]
        dh, dw = im.out_hw
        if min(sh, sw, dh, dw) <= 0:
            raise ValueError(f'empty image in the batch: {im.raw.shape} -> {im.out_hw}')
        row = np.zeros(12, dtype=np.int32)
        row[:6] = [sh, sw, dh, dw, _FLIP_BITS[im.flip], int(sh == 2 * dh and sw == 2 * dw)]
        row[8:].view(np.float64)[:] = [1.0 / (float(dw) / float(sw)), 1.0 / (float(dh) / float(sh))]
        metas.append(row)
        offs.append(total)
        total += (im.raw.size + 15) // 16 * 16
    Hp = max((im.pad_hw or im.out_hw)[0] for im in images)
    Wp = max((im.pad_hw or im.out_hw)[1] for im in images)
    return norm, pad_val, np.stack(metas), np.asarray(offs, dtype=np.int64), total, Hp, Wp


class DeviceBatchStager:
    """Reusable pinned staging buffer + the launch.  One H2D copy per batch: [offsets | meta | pixels]."""

    def __init__(self, device='cuda:0'):
        self.device = torch.device(device)
        self._pinned = None
        self._in_flight = None                 # event after the last H2D copy out of the staging buffer

    def _staging(self, nbytes):
        if self._in_flight is not None:
            self._in_flight.synchronize()      # the previous batch has left the pinned buffer
        if self._pinned is None or self._pinned.numel() < nbytes:
            self._pinned = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        return self._pinned[:nbytes]

    def upload(self, images):
        """-> (packed device bytes, plan) with the raw pixels of `images` resident in HBM."""
        norm, pad_val, meta, offs, total, Hp, Wp = _plan(images)
        B = len(images)
        head = (B * 8 + B * 48 + 15) // 16 * 16
        host = self._staging(head + total)
        hv = host.numpy()
        hv[:B * 8] = offs.view(np.uint8)
        hv[B * 8:B * 8 + B * 48] = meta.reshape(-1).view(np.uint8)
        for im, o in zip(images, offs):
            hv[head + o:head + o + im.raw.size] = im.raw.reshape(-1)
        dev = host.to(self.device, non_blocking=True)
        self._in_flight = torch.cuda.Event()
        self._in_flight.record()
        plan = dict(B=B, Hp=Hp, Wp=Wp, head=head, norm=norm, pad_val=pad_val)
        return dev, plan

    def run(self, dev, plan):
        """Launch the fused kernel on the current stream -> (B, 3, Hp, Wp) fp32, channels_last."""
        B, Hp, Wp, head = plan['B'], plan['Hp'], plan['Wp'], plan['head']
        mean, std, to_rgb = plan['norm']
        out = torch.empty((B, Hp, Wp, 3), device=dev.device, dtype=torch.float32)
        base = dev.data_ptr()
        import ctypes
        capi.call('htd_image_batch_pipeline', ctypes.c_void_p(base + head), ctypes.c_void_p(base),
                  ctypes.c_void_p(base + B * 8), capi.ptr(out), B, Hp, Wp, float(mean[0]), float(mean[1]),
                  float(mean[2]), float(std[0]), float(std[1]), float(std[2]), int(to_rgb), float(plan['pad_val']),
                  capi.current_stream_ptr(), work=('byte', 12.0 * B * Hp * Wp, 12.0 * B * Hp * Wp))
        return out.permute(0, 3, 1, 2)

    def __call__(self, images):
        dev, plan = self.upload(images)
        return self.run(dev, plan)


_STAGERS = {}


def collate(samples, device='cuda:0'):
    """mmcv.parallel.collate + scatter for one GPU's samples (the dicts `Collect` returns): the image batch is produced
    on `device` by the fused kernel, boxes / labels become per-image device tensors, metas stay host dicts --
    exactly the keyword arguments of TwoStageDetector.forward_train / simple_test."""
    device = torch.device(device)
    stager = _STAGERS.setdefault(device, DeviceBatchStager(device))
    if isinstance(samples[0]['img'], list):                              # MultiScaleFlipAug: one batch per augmentation
        n_aug = len(samples[0]['img'])
        return dict(img=[stager([s['img'][a] for s in samples]) for a in range(n_aug)],
                    img_metas=[[s['img_metas'][a] for s in samples] for a in range(n_aug)])
    out = dict(img=stager([s['img'] for s in samples]), img_metas=[s['img_metas'] for s in samples])
    for key in samples[0]:
        if key not in ('img', 'img_metas'):
            out[key] = [s[key].to(device, non_blocking=True) if torch.is_tensor(s[key]) else s[key] for s in samples]
    return out
