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
