"""Checkpoint I/O in the reference's wire format.

The reference loads and saves through mmcv's `load_checkpoint` / `save_checkpoint` (mmcv-full 1.2.1, absent third
party; call sites backbones/resnet.py:598-600, apis/train.py:146-149, tools/test.py): a pickled dict
`{'meta': ..., 'state_dict': OrderedDict(name -> tensor), ['optimizer': ...]}`, keys optionally prefixed by
`module.` (saved from a DataParallel wrapper), tensors in PyTorch's logical layouts.  This module reads and writes
exactly that, so the authors' released `.pth` files and torchvision ResNet weights load unchanged:
  * parameter names are the reference's (the module tree mirrors it), so no key mapping is needed;
  * convolution weights live in KRSC (channels_last) memory and the first RoI-head FC in (out,h,w,C) order --
    `load_state_dict` / `state_dict` convert through strides and TileLinear's hooks, the files keep logical shapes;
  * frozen BatchNorm statistics are ordinary buffers (the fold into the convolution happens at run time).
Only local files: there is no network on this path (`torchvision://`, `open-mmlab://`, `http(s)://` raise).
"""
import os
import time
from collections import OrderedDict

import torch

_REMOTE = ('modelzoo://', 'torchvision://', 'open-mmlab://', 'openmmlab://', 'mmcls://', 'http://', 'https://', 's3://')


def _strip_prefix(state_dict, prefix='module.'):
    if state_dict and all(k.startswith(prefix) for k in state_dict):
        return OrderedDict((k[len(prefix):], v) for k, v in state_dict.items())
    return state_dict


def load_state_dict(module, state_dict, strict=False, logger=None):
    """mmcv.runner.load_state_dict semantics: copy what matches, collect what does not, raise only when strict.
    -> (missing_keys, unexpected_keys, size_mismatches)."""
    own = module.state_dict()
    mismatched = []
    usable = OrderedDict()
    for k, v in state_dict.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            mismatched.append((k, tuple(v.shape), tuple(own[k].shape)))
            continue
        usable[k] = v
    res = module.load_state_dict(usable, strict=False)
    missing = [k for k in res.missing_keys if 'num_batches_tracked' not in k]
    unexpected = list(res.unexpected_keys)
    msgs = []
    if unexpected:
        msgs.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        msgs.append('missing keys in source state_dict: ' + ', '.join(missing))
    for k, a, b in mismatched:
        msgs.append(f'size mismatch for {k}: checkpoint {a} vs model {b}')
    if msgs:
        text = 'The model and loaded state dict do not match exactly\n' + '\n'.join(msgs)
        if strict:
            raise RuntimeError(text)
        (logger.warning if logger is not None else print)(text)
    return missing, unexpected, mismatched


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    """Load `filename` (a local .pth in the reference format, or a bare state_dict) into `model`; returns the
    checkpoint dict (with 'meta' when present), like mmcv.runner.load_checkpoint."""
    if filename.startswith(_REMOTE):
        raise IOError(f'{filename}: remote checkpoints are not available here (no network); download the file and '
                      'pass its local path')
    if not os.path.isfile(filename):
        raise IOError(f'{filename} is not a checkpoint file')
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(checkpoint, dict) and 'state_dict' in checkpoint:
        state_dict = checkpoint['state_dict']
    elif isinstance(checkpoint, dict) and 'model' in checkpoint and isinstance(checkpoint['model'], dict):
        state_dict = checkpoint['model']
    elif isinstance(checkpoint, dict):
        state_dict = checkpoint
    else:
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    load_state_dict(model, _strip_prefix(OrderedDict(state_dict)), strict, logger)
    return checkpoint if isinstance(checkpoint, dict) else dict(state_dict=state_dict)


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """Write the reference's format: {'meta', 'state_dict' (CPU tensors, logical layouts) [, 'optimizer']}."""
    meta = dict(meta or {})
    meta.setdefault('time', time.asctime())
    if hasattr(model, 'module'):
        model = model.module
    state = OrderedDict((k, v.detach().cpu().contiguous()) for k, v in model.state_dict().items())
    ckpt = dict(meta=meta, state_dict=state)
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict() if hasattr(optimizer, 'state_dict') else optimizer
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(ckpt, filename)
    return filename
