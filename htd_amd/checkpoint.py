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
Only local files: there is no network on this path.  `torchvision://` / `open-mmlab://` names (the reference configs'
`pretrained`) resolve to files in $HTD_PRETRAINED_DIR, `http(s)://` raises.
"""
import os
import pickle
import time
from collections import OrderedDict

import torch

_REMOTE = ('modelzoo://', 'torchvision://', 'open-mmlab://', 'openmmlab://', 'mmcls://', 'http://', 'https://', 's3://')


def _strip_prefix(state_dict, prefix='module.'):
    """mmcv 1.2.1 decides on the FIRST key (`if list(state_dict.keys())[0].startswith('module.')`) and then cuts
    len(prefix) characters off every key."""
    keys = list(state_dict.keys())
    if keys and keys[0].startswith(prefix):
        return OrderedDict((k[len(prefix):], v) for k, v in state_dict.items())
    return state_dict


def _resolve(filename):
    """Local path for a checkpoint name.  `torchvision://resnet50`, `open-mmlab://...` (what the reference configs
    put in `pretrained`, configs/htd/htd_resnet50_1x.py:7) are looked up as <name>.pth / <name>-*.pth in the directory
    $HTD_PRETRAINED_DIR -- there is no network on this path, so the files have to be placed there."""
    if filename.startswith(_REMOTE):
        scheme, name = filename.split('://', 1)
        cache = os.environ.get('HTD_PRETRAINED_DIR')
        if scheme in ('torchvision', 'open-mmlab', 'openmmlab', 'modelzoo', 'mmcls') and cache and os.path.isdir(cache):
            base = name.replace('/', '_')
            for f in sorted(os.listdir(cache)):
                if f == base + '.pth' or (f.startswith(base + '-') and f.endswith('.pth')):
                    return os.path.join(cache, f)
        raise IOError(f'{filename}: remote checkpoints are not available here (no network); download the file and pass '
                      f'its local path, or put it as {name.replace("/", "_")}.pth into the directory named by HTD_PRETRAINED_DIR')
    return filename


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


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None, trusted=None):
    """Load `filename` (a local .pth in the reference format, or a bare state_dict) into `model`; returns the
    checkpoint dict (with 'meta' when present), like mmcv.runner.load_checkpoint.

    Files are read with torch.load(weights_only=True): tensors, containers and plain scalars only -- enough for every
    file this package writes and for torchvision / mmdet weight files.  A checkpoint that pickles other objects in its
    'meta' needs trusted=True (or HTD_TRUST_CHECKPOINTS=1): unpickling runs code chosen by whoever wrote the file."""
    filename = _resolve(filename)
    if not os.path.isfile(filename):
        raise IOError(f'{filename} is not a checkpoint file')
    if trusted is None:
        trusted = os.environ.get('HTD_TRUST_CHECKPOINTS') == '1'
    try:
        checkpoint = torch.load(filename, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        # ONLY the weights-only refusal leads to the trusted path; a truncated / corrupt / mistyped file raises its own error
        # (zip, EOF, map_location ...) and must not nudge anyone towards arbitrary-code unpickling
        if not trusted:
            raise RuntimeError(f'{filename} holds pickled objects beyond tensors and plain containers ({e}); pass '
                               'trusted=True / set HTD_TRUST_CHECKPOINTS=1 only for files from a source you trust') from e
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
        opt = optimizer.state_dict() if hasattr(optimizer, 'state_dict') else optimizer
        ckpt['optimizer'] = dict(opt, state={k: {n: (t.detach().cpu() if torch.is_tensor(t) else t) for n, t in st.items()}
                                             for k, st in opt.get('state', {}).items()})
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(ckpt, filename)
    return filename
