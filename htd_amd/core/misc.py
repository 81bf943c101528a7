"""mmdet/core/utils/misc.py:7-39."""
from functools import partial

import torch


def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def unmap(data, count, inds, fill=0):
    """Scatter `data` (rows where inds is true) back into a `count`-row tensor filled with `fill`."""
    ret = data.new_full((count, ) + tuple(data.shape[1:]), fill)
    ret[inds.type(torch.bool)] = data
    return ret


_CONSTS = {}


def const_tensor(values, device, dtype=torch.float32):
    """Device-resident constant built once per (values, device, dtype): a fresh torch.tensor(list, device=gpu)
    is a pageable host-to-device copy, i.e. a full host/device synchronisation on every call."""
    def freeze(v):
        return tuple(freeze(u) for u in v) if isinstance(v, (list, tuple)) else v
    key = (freeze(values), str(device), dtype)
    t = _CONSTS.get(key)
    if t is None:
        if len(_CONSTS) > 4096:
            _CONSTS.clear()
        t = _CONSTS[key] = torch.tensor(values, dtype=dtype, device=device)
    return t


_ARANGES = {}


def arange_cached(n, device, dtype=torch.int64, start=0):
    """torch.arange(start, start + n) built once per (n, device, dtype, start): READ-ONLY (the step builds the same index
    ramps every iteration; each one is a kernel launch otherwise)."""
    key = (int(n), str(device), dtype, int(start))
    t = _ARANGES.get(key)
    if t is None:
        if len(_ARANGES) > 512:
            _ARANGES.clear()
        t = _ARANGES[key] = torch.arange(start, start + n, device=device, dtype=dtype)
    return t
