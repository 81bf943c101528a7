"""Dense contractions of the path (convolutions, Linear layers) -- the MFMA-bound part.

`conv2d` / `linear` are the single entry points every module uses.  They run the hand-written fp32
matrix-core implicit-GEMM kernels of libhtd_amd.so (htd_conv2d_fwd / _bwd_data / _bwd_weight,
csrc/conv_fwd.hip, csrc/conv_wgrad.hip) through the C ABI.  Activations NHWC, weights KRSC; the forward epilogue
carries bias, residual add (optionally through nearest up-sampling) and ReLU; in backward the ReLU masks and residual
joins live in the data-gradient epilogues (ResStageFunction), bias gradients come out of the weight-gradient launch
and parameter gradients are written straight into the flat gradient buffer (gradient sinks).
GPU tensors only -- there is no other path.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.multiprocessing.reductions import StorageWeakRef as _StorageWeakRef

from . import capi

CL = torch.channels_last
_P = capi.ptr
_S = capi.current_stream_ptr


def _out_hw(H, W, kh, kw, stride, pad, dil):
    return ((H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1, (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1)


def _need_gpu(t, name):
    if not t.is_cuda:
        raise NotImplementedError(f'{name}: only GPU tensors are supported (libhtd_amd.so has no CPU path)')


def _splitk_ws(M, Co, Cred, kh, kw, device):
    """Split-K scratch for a GEMM with M rows, Co columns, reduction kh*kw*Cred (None when not worth splitting)."""
    nbytes = capi.lib().htd_conv2d_workspace_bytes(M, Co, Cred, kh, kw)
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32) if nbytes > 0 else None


# ---- gradient sinks ------------------------------------------------------------------------------------------
# runner.FlatParams registers, for every leaf parameter, its slice of the flat gradient buffer.  A kernel that
# produces the gradient of such a parameter writes it straight into that slice and returns an alias of it, so
# autograd's AccumulateGrad adopts the tensor instead of launching `grad += new` once per parameter (~230 launches
# per step).  A slice is handed out once per step; further gradients of the same parameter (shared RPN convs, the
# stage-1 classifier reused by stage 2) use fresh tensors, autograd sums them, and FlatParams.collect() copies any
# gradient that did not end up in its slice.
# The registry is keyed by the parameter's address but every entry carries a weak reference to the Parameter and
# belongs to one FlatParams (its own "used this step" set): several trainers can coexist, a dropped trainer's entries
# disappear with it (FlatParams.close / __del__), and an entry whose parameter died or moved is never handed out to
# a later tensor that happens to be allocated at the same address.
_GRAD_SINK = {}                 # data_ptr -> (slice view, weakref to the Parameter, owner's used-set)


def register_grad_sinks(params, views, used):
    """params[i]'s gradient slice is views[i]; `used` is the owner's per-step set.  -> the keys registered."""
    import weakref
    keys = []
    for p, v in zip(params, views):
        _GRAD_SINK[p.data_ptr()] = (v, weakref.ref(p), used)
        keys.append(p.data_ptr())
    used.clear()
    return keys


def unregister_grad_sinks(keys, used):
    for k in keys:
        ent = _GRAD_SINK.get(k)
        if ent is not None and ent[2] is used:
            del _GRAD_SINK[k]


_TEMP_USED = set()
_TEMP_KEYS = []


def temp_grad_sink(t):
    """A gradient buffer for the NON-LEAF tensor t that several layers read (the RPN's merged 1x1 head weights, made once per
    step and used on five pyramid levels): the first weight gradient of the step is written into it and returned to autograd,
    the later ones are added in place by the kernel and return nothing (_wgrad_launch) -- one defined contribution instead of
    five tensors to sum.  Dropped by new_step()."""
    import weakref
    if not (t.is_cuda and t.requires_grad and SINK_ACCUMULATE):
        return
    _GRAD_SINK[t.data_ptr()] = (torch.empty_like(t), weakref.ref(t), _TEMP_USED)
    _TEMP_USED.discard(t.data_ptr())
    _TEMP_KEYS.append(t.data_ptr())


def _drop_temp_sinks():
    for k in _TEMP_KEYS:
        ent = _GRAD_SINK.get(k)
        if ent is not None and ent[2] is _TEMP_USED:
            del _GRAD_SINK[k]
    _TEMP_KEYS.clear()
    _TEMP_USED.clear()


def reset_grad_sinks(used=None):
    if used is not None:
        used.clear()
    else:
        for ent in _GRAD_SINK.values():
            ent[2].clear()
    _SIDE_CONSUMED.clear()


def grad_out(like):
    """Tensor to write the gradient of `like` into: an alias of its registered flat-gradient slice, else a new one."""
    return grad_out2(like)[0]


def grad_sink_again(like):
    """The registered flat-gradient slice of `like` when it was ALREADY handed out this step (a parameter several layers share),
    viewed in like's layout -- for a kernel that adds into it (htd_conv2d_bwd_weight_acc); else None."""
    key = like.data_ptr()
    ent = _GRAD_SINK.get(key)
    if ent is None or torch.is_grad_enabled():
        return None
    v, ref, used = ent
    owner = ref()
    if owner is None or owner.data_ptr() != key or key not in used or v.numel() != like.numel():
        return None
    if v.shape == like.shape and v.stride() == like.stride():
        return v.view_as(v)
    if like.dim() == 4 and like.shape[2:] == (1, 1) and like.size(0) == v.size(0):
        if v.dim() == 4 and v.is_contiguous(memory_format=CL):
            v = v.permute(0, 2, 3, 1).reshape(v.size(0), -1)
        if v.dim() == 2 and v.is_contiguous():
            return v.view(v.size(0), 1, 1, v.size(1)).permute(0, 3, 1, 2)
    return None


def grad_out2(like):
    """-> (tensor, is_sink)."""
    key = like.data_ptr()
    ent = _GRAD_SINK.get(key)
    if ent is not None:
        v, ref, used = ent
        owner = ref()
        if owner is None or owner.data_ptr() != key:          # the parameter is gone or was re-pointed: stale entry
            del _GRAD_SINK[key]
            ent = None
    if ent is not None and key not in used and v.numel() == like.numel() and torch.is_grad_enabled() is False:
        if v.shape == like.shape and v.stride() == like.stride():
            used.add(key)
            # (a temporary sink's buffer is read by autograd on the MAIN stream -- CatBackward of the merged RPN head weights --
            #  so it does not count as a flat-buffer sink for the weight-gradient stream's "nobody waits" rule, _wgrad_raw)
            return v.view_as(v), used is not _TEMP_USED
        if like.dim() == 4 and like.shape[2:] == (1, 1) and like.size(0) == v.size(0):
            if v.dim() == 4 and v.is_contiguous(memory_format=CL):       # TileLinear: (out,C,h,w) stored (out,h,w,C)
                v = v.permute(0, 2, 3, 1).reshape(v.size(0), -1)
            if v.dim() == 2 and v.is_contiguous():
                used.add(key)                         # Linear weight seen as a 1x1 conv (channels_last view)
                return v.view(v.size(0), 1, 1, v.size(1)).permute(0, 3, 1, 2), True
    if like.dim() == 4:
        return torch.empty(like.shape, device=like.device, dtype=like.dtype, memory_format=CL), False
    return torch.empty_like(like), False


# ---- weight-gradient stream --------------------------------------------------------------------------------------
# In backward the data gradients form the critical chain; weight gradients (wgrad -> split-K sum -> BN-fold gradient ->
# flat gradient buffer) are only needed by the optimizer.  They are queued on a second HIP stream: their small reduce /
# fold kernels and the tail waves of the big ones then overlap with the data-gradient kernels of the main stream
# instead of leaving CUs idle.  Rules: the side stream waits for the main stream before every launch (operands ready);
# a result that autograd will consume on the main stream (not a flat-buffer sink, not an input of a BN fold that itself
# runs on the side stream) makes the main stream wait at once; runner joins the streams after backward.
# Measured (HTD-R50, B=4): round 1 75.2 -> 74.0 ms per step, round 3 91.2 -> 92.5..93.2 img/s.  Every overlapped kernel's own
# duration grows (they share the CUs), which would blur the per-kernel roofline timing of bench.py: on the steps whose calls
# are bracketed by device events (capi.profiling(): every 4th step of the timed region) the weight gradients stay on the
# main stream, so a kernel's measured duration is its duration alone on the chip.  HTD_OVERLAP_WGRAD=0 switches it off.
OVERLAP_WGRAD = bool(int(__import__('os').environ.get('HTD_OVERLAP_WGRAD', '1')))
_OVERLAP_IN_PROFILE = False
_SIDE = {}
_SIDE_CONSUMED = set()          # data_ptr of folded weights whose gradient is consumed by _BNFold.backward (side stream)


class overlap_wgrad:
    """`with overlap_wgrad(flag):` -- the weight-gradient stream on / off for the steps of ONE model (runner.Trainer reads
    the model's `overlap_wgrad` attribute; None keeps the process default), instead of a module global that the
    construction of a bf16 model would leave switched off for every model built after it."""

    def __init__(self, flag):
        self.flag = flag

    def __enter__(self):
        global OVERLAP_WGRAD
        self.saved = OVERLAP_WGRAD
        if self.flag is not None:
            OVERLAP_WGRAD = bool(self.flag)

    def __exit__(self, *exc):
        global OVERLAP_WGRAD
        OVERLAP_WGRAD = self.saved


def side_stream(device=None):
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    s = _SIDE.get(dev)
    if s is None:
        s = _SIDE[dev] = torch.cuda.Stream(device=dev)
    return s


def mark_side_consumed(t):
    _SIDE_CONSUMED.add(t.data_ptr())


# Flipped / transposed weight images produced ahead of time (bricks._BNFoldMany writes them with the folds of a whole
# stage): offered under the folded weight's address during the forward pass, taken -- and removed -- by the autograd
# node that will run the data gradient (ResStageFunction.forward), which keeps them with its saved tensors.
_FLIPPED = {}


def offer_flipped(weight, wT):
    _FLIPPED[weight.data_ptr()] = wT


def take_flipped(weight):
    return _FLIPPED.pop(weight.data_ptr(), None)


# Flipped / transposed weight images made during this step, by (address, version, shape, parameter epoch): a weight that several data gradients
# read (the RPN's shared convolutions, five levels) is flipped once, and flip_many() makes the images of all plain
# convolutions of the step in one launch.  new_step() drops them (the optimizer rewrites the weights in place).  Every entry
# holds its weight tensor too: while the entry lives the allocator cannot hand that address to another weight.
_STEP_FLIPS = {}


_M = None


def _flip_key(w):
    # mmcv_ops.PARAM_EPOCH: the optimizer kernel and the rank-0 broadcast rewrite weights without touching tensor._version
    global _M
    if _M is None:
        from . import mmcv_ops
        _M = mmcv_ops
    return (w.data_ptr(), w._version, tuple(w.shape), _M.PARAM_EPOCH)


def new_step():
    _STEP_FLIPS.clear()
    _STEP_PLANES.clear()
    _AMAX_POOL.clear()
    _AMAX_BY_PTR.clear()
    _drop_temp_sinks()


# ---- H2 arithmetic of the 3x3 layers (csrc/conv_x3.hip, h2_scale): the kernel scales its activation operand by a power of two
# taken from the tensor's largest magnitude, a device scalar that htd_absmax leaves in a slot of a per-step pool of zeros
# (one fill per step, no host read).  Never cached per tensor: the allocator hands the same address to different tensors.
_AMAX_POOL = {}
_AMAX_SLOTS = 1024
# The maximum of a layer's input normally comes for free -- left behind by the epilogue that wrote the tensor (X3Params::amax_out,
# carried on the tensor object as `_htd_amax`); where it does not, see H2_ABSMAX_MIN_WORK.  HTD_H2_1X1=0: 3x3 layers only.
H2_1X1 = os.environ.get('HTD_H2_1X1', '1') != '0'
H2_CHECK = os.environ.get('HTD_H2_CHECK', '0') == '1'        # tests: every carried maximum is compared with a fresh htd_absmax
# A tensor that carries no maximum gets a pass of htd_absmax when every element of it goes into at least this many multiply-adds
# of the launch (filter taps x the OTHER side's channels); below it the layer runs on the three-piece bf16 form.  Measured on the
# train step (headline / trained-like ms, two alternating runs each): never 31.70 / 43.4, from 512 on (1x1 into >= 512 channels,
# the FC layers) 32.33 / 41.7, from 2304 on (the 3x3 layers of >= 256 channels) 31.68 / 40.9.
H2_ABSMAX_MIN_WORK = int(os.environ.get('HTD_H2_ABSMAX_MIN_WORK', '2304'))
# ... and only for tensors of at least this many elements: below it the pass is a launch (8 us and a slot in the queue), not bytes
H2_ABSMAX_MIN_ELEMS = int(os.environ.get('HTD_H2_ABSMAX_MIN_ELEMS', str(1 << 20)))


def _x3h_ok(Cred, Cout, kh, kw, stride, padding, dilation, dtype):
    return dtype == torch.float32 and (kh == 3 or H2_1X1) and \
        bool(capi.lib().htd_conv2d_x3h_supported(Cred, Cout, kh, kw, stride, padding, dilation))


def _amax_slot(device):
    ent = _AMAX_POOL.get(device.index)
    if ent is None or ent[1] >= _AMAX_SLOTS:
        ent = [torch.zeros(_AMAX_SLOTS, device=device, dtype=torch.float32), 0]
        _AMAX_POOL[device.index] = ent
    slot = ent[0][ent[1]:ent[1] + 1]
    ent[1] += 1
    return slot


def absmax(x):
    """-> device scalar (1-element view) holding max |x| (x dense fp32 on the GPU)."""
    slot = _amax_slot(x.device)
    capi.call('htd_absmax', _P(x), x.numel(), _P(slot), _S(), work=('byte', 4.0 * x.numel()))
    return slot


# HTD_H2_GUARD=1 (and the check mode): every H2 launch is remembered as (maximum it was given, maximum its epilogue left), and
# dense.h2_check() -- a device read -- raises when an output maximum is NaN / inf although the input's was finite: the input
# maximum was not the tensor's (a stale `_htd_amax`), the split overflowed fp16.  Off by default (an overflow is not silent
# anyway: infinities in the output, NaN in the loss).
H2_GUARD = H2_CHECK or os.environ.get('HTD_H2_GUARD', '0') == '1'
_H2_LAUNCHES = []


def _h2_guard(am_in, out_slot):
    if H2_GUARD and out_slot is not None:
        _H2_LAUNCHES.append((am_in, out_slot))


def h2_check(device=None):
    """Raise if an H2 launch since the last call (guard mode) overflowed: see H2_GUARD.  Reads the device."""
    pairs, _H2_LAUNCHES[:] = list(_H2_LAUNCHES), []
    if not pairs:
        return
    a = torch.cat([p[0].reshape(1) for p in pairs])
    b = torch.cat([p[1].reshape(1) for p in pairs])
    bad = torch.isfinite(a) & ~torch.isfinite(b)
    if bool(bad.any()):
        i = int(torch.nonzero(bad)[0])
        raise RuntimeError(f'htd_amd: H2 convolution launch {i} of {len(pairs)} was given a maximum ({float(a[i])}) smaller than its input '
                           'tensor holds (stale _htd_amax): its output contains infinities')


H2_TRACE = {} if os.environ.get('HTD_H2_TRACE', '0') == '1' else None      # diagnostics: which 1x1 launches found no carried maximum


def _h2_trace(what, t, weight):
    if H2_TRACE is not None:
        import traceback
        fr = [f for f in traceback.extract_stack(limit=12) if 'dense.py' not in f.filename]
        key = (what, tuple(t.shape), tuple(weight.shape), f'{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}' if fr else '?',
               'grad_fn' if t.grad_fn is not None else ('attr-stale' if hasattr(t, '_htd_amax') else 'no-attr'))
        H2_TRACE[key] = H2_TRACE.get(key, 0) + 1


# A tagged tensor reaches its consumer as ANOTHER tensor object when views lie between them (linear()'s (M, K) <-> (M, K, 1, 1), the
# view nodes autograd runs on the way back, a saved tensor unpacked in backward): the maxima are therefore also kept by ADDRESS,
# each with a weak reference to the storage it was written into -- while that storage lives the allocator cannot hand the address to
# anybody else, and a dense tensor of the same address, element count and version counter over the same storage holds the same
# elements.  HTD_H2_VIEWS=0: tensor objects only.
H2_VIEWS = os.environ.get('HTD_H2_VIEWS', '1') != '0'
_AMAX_BY_PTR = {}


def _dense_layout(t):
    """Does t cover numel() distinct consecutive elements from data_ptr() on (a permutation of a contiguous tensor)?"""
    if t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=CL)):
        return True
    want = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz != 1), key=lambda p: p[1]):
        if stride != want:
            return False
        want *= size
    return True


def tag_amax(t, slot):
    """Remember that `slot` holds max |t| (written by the kernel that is filling t): on the tensor object, and by address."""
    t._htd_amax = (slot, t.data_ptr(), t.numel(), t._version)
    if H2_VIEWS:
        st = t.untyped_storage()
        _AMAX_BY_PTR[t.data_ptr()] = (slot, t.numel(), t._version, st._cdata, _StorageWeakRef(st))


def drop_amax(t):
    """t is about to be modified in place by a kernel torch does not see: its carried maximum is void."""
    if t is not None:
        if hasattr(t, '_htd_amax'):
            del t._htd_amax
        _AMAX_BY_PTR.pop(t.data_ptr(), None)


def carried_amax(t):
    a = getattr(t, '_htd_amax', None)
    if a is None and H2_VIEWS:
        e = _AMAX_BY_PTR.get(t.data_ptr())
        if e is not None and e[1] == t.numel() and e[2] == t._version and not e[4].expired() and \
                t.untyped_storage()._cdata == e[3] and _dense_layout(t):
            a = (e[0], t.data_ptr(), e[1], e[2])
    if a is None or a[1] != t.data_ptr() or a[2] != t.numel() or a[3] != t._version:
        return None
    if H2_CHECK:
        fresh = absmax(t)
        if not torch.equal(fresh, a[0]) and not (torch.isnan(fresh).item() and torch.isnan(a[0]).item()):
            raise RuntimeError(f'stale carried maximum: {float(a[0])} against {float(fresh)} on a tensor of shape {tuple(t.shape)}')
    return a[0]


# bf16 plane images of the weights (csrc/conv_x3.hip: the B operand of conv_x3p_kernel, split once per step instead of
# once per tile and tap).  Keyed like _STEP_FLIPS plus the parameter epoch of mmcv_ops (the optimizer kernel and the rank-0
# broadcast rewrite weights without touching tensor._version); every entry keeps its weight alive; new_step() drops them.
_STEP_PLANES = {}


def _planes_key(w, transposed):
    return _flip_key(w) + (bool(transposed), )


_ALSO_WANTED = set()          # (weight shape, transposed, h2): images that a step asked for outside its table launch (by shape: the
                              # folded weights of a stage live at another address every step)


def _planes_h2(w, transposed):
    """Is the image the step's table launch makes for this weight the H2 one (two fp16 pieces + row scales)?  The layers that
    run on H2 whenever they can; the other image of a weight is made on demand (x3_planes(..., h2=...))."""
    Co, Ci, kh, kw = w.shape
    cred, cout = (Co, Ci) if transposed else (Ci, Co)
    return kh == kw and _x3h_ok(cred, cout, kh, kw, 1, kh // 2, 1, w.dtype)


def x3_planes(weight, transposed, h2=None):
    """weight (Co,Ci,kh,kw) channels_last fp32 -> its plane image (forward operand, or data-gradient operand when
    `transposed`), made once per step.  h2: the H2 image (None: whatever the table launch makes for this weight)."""
    if h2 is None:
        h2 = _planes_h2(weight, transposed)
    key = _planes_key(weight, transposed) + (bool(h2), )
    hit = _STEP_PLANES.get(key)
    if hit is not None:
        return hit[1]
    # made on demand, one launch (two for H2) for this weight alone: remember the request, the next steps' table launch
    # (planes_many) makes this image too
    _ALSO_WANTED.add((tuple(weight.shape), bool(transposed), bool(h2)))
    Co, Ci, kh, kw = weight.shape
    nbytes = capi.lib().htd_conv2d_x3_planes_bytes(Co, kh, kw, Ci, int(transposed))
    planes = torch.empty(nbytes // 4, device=weight.device, dtype=torch.int32)
    capi.call('htd_conv2d_x3h_planes' if h2 else 'htd_conv2d_x3_planes', _P(weight), _P(planes), Co, kh, kw, Ci, int(transposed), _S())
    _STEP_PLANES[key] = (weight, planes)
    return planes


def _planes_wanted(w, transposed):
    """Could conv_x3p_kernel take a layer with this weight as its forward (transposed: data-gradient) operand?  Shape test only --
    stride / padding are not known here; a 3x3 s2 layer's unused image costs one pass over its weights."""
    Co, Ci, kh, kw = w.shape
    cred, cout = (Co, Ci) if transposed else (Ci, Co)
    return w.is_cuda and w.dtype == torch.float32 and kh == kw and kh in (1, 3) and w.is_contiguous(memory_format=CL) and \
        bool(capi.lib().htd_conv2d_x3p_supported(cred, cout, kh, kw, 1, kh // 2, 1))


def planes_many(items):
    """items: [(weight (Co,Ci,kh,kw) channels_last fp32, transposed)] -> their plane images in ONE launch
    (htd_conv2d_x3_planes_many), registered for x3_planes() to find.  Entries already made this step are skipped."""
    import numpy as np
    todo, seen = [], set()
    for w, tr in items:
        w4 = w if w.dim() == 4 else w.view(w.size(0), w.size(1), 1, 1)
        if not _planes_wanted(w4, tr):
            continue
        h2 = _planes_h2(w4, tr)
        for v in ((h2, ) if (tuple(w4.shape), bool(tr), not h2) not in _ALSO_WANTED else (h2, not h2)):
            key = _planes_key(w4, tr) + (v, )
            if key not in _STEP_PLANES and key not in seen:
                seen.add(key)
                todo.append((key, w4, tr))
    if not todo:
        return
    L = capi.lib()
    dev = todo[0][1].device
    for h2 in (False, True):                 # the bf16 images in one launch, the H2 images (row scales, then planes) in two
        part = [t for t in todo if t[0][-1] == h2]
        if not part:
            continue
        sizes = [L.htd_conv2d_x3_planes_bytes(w.size(0), w.size(2), w.size(3), w.size(1), int(tr)) // 4 for _, w, tr in part]
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.int32)
        desc = np.zeros((len(part), 6 if h2 else 5), dtype=np.int64)
        off = block0 = row0 = 0
        for i, ((key, w, tr), n32) in enumerate(zip(part, sizes)):
            out = flat[off:off + n32]
            off += n32
            Co, Ci, taps = w.size(0), w.size(1), w.size(2) * w.size(3)
            N = Ci if tr else Co
            Np = (N + 127) // 128 * 128
            desc[i, 0], desc[i, 1] = w.data_ptr(), out.data_ptr()
            desc[i, 2] = Co | (taps << 32)                  # two int32 per int64 slot (little endian)
            desc[i, 3] = Ci | (int(tr) << 32)
            desc[i, 4] = block0
            K = Co if tr else Ci
            nch = (taps * K + 1023) // 1024                               # chunks of the row-maximum pass (csrc: H2_ECH)
            if h2:
                desc[i, 5] = row0
                row0 += Np // 64 * nch
            block0 += ((n32 - (2 + nch) * Np) // 4 // 6 + 255) // 256     # elements = uint4 count of the planes / 6 chunks
            _STEP_PLANES[key] = (w, out)
        table = capi.upload_table(desc, dev)
        if h2:
            capi.call('htd_conv2d_x3h_planes_many', _P(table), len(part), block0, row0, _S())
        else:
            capi.call('htd_conv2d_x3_planes_many', _P(table), len(part), block0, _S())


def _x3p_ok(Cred, Cout, kh, kw, stride, padding, dilation, dtype):
    return dtype == torch.float32 and bool(capi.lib().htd_conv2d_x3p_supported(Cred, Cout, kh, kw, stride, padding, dilation))


def flip_many(weights):
    """One launch (htd_bn_fold_many_fwd without BN entries) writing the dgrad images of `weights` (Co,Ci,kh,kw)
    channels_last or (Co,Ci) -- picked up by _dgrad_raw through _STEP_FLIPS."""
    import numpy as np
    ws, pl = [], []
    for w in weights:
        w4 = w if w.dim() == 4 else w.view(w.size(0), w.size(1), 1, 1)
        pl += [(w4, False), (w4, True)]
        if w4.is_cuda and w4.dtype == torch.float32 and w4.size(0) % 8 == 0 and w4.is_contiguous(memory_format=CL) and \
                _flip_key(w4) not in _STEP_FLIPS and not _planes_wanted(w4, True):
            ws.append(w4)                   # data gradient on conv_igemm_kernel: needs the fp32 flipped image
    planes_many(pl)                         # conv_x3p_kernel operands (forward and data gradient) of every layer it takes
    if not ws:
        return
    dev = ws[0].device
    flat = torch.empty(sum(w.numel() for w in ws), device=dev, dtype=torch.float32)
    desc = np.zeros((len(ws), 10), dtype=np.int64)
    off = tile0 = 0
    for i, w in enumerate(ws):
        Co, Ci, taps = w.size(0), w.size(1), w.size(2) * w.size(3)
        wT = flat[off:off + w.numel()]
        off += w.numel()
        desc[i, 0], desc[i, 7] = w.data_ptr(), wT.data_ptr()
        desc[i, 8] = Co | (Ci << 32)
        desc[i, 9] = taps | (tile0 << 32)
        tile0 += taps * ((Co + 31) // 32) * ((Ci + 31) // 32)
        _STEP_FLIPS[_flip_key(w)] = (w, wT)
    table = capi.upload_table(desc, dev)
    capi.call('htd_bn_fold_many_fwd', _P(table), len(ws), tile0, 0.0, _S())


def join_side_stream():
    """Main stream waits for everything queued on the weight-gradient stream (call after backward)."""
    if _SIDE:
        cur = torch.cuda.current_stream()
        s = _SIDE.get(cur.device.index)
        if s is not None:
            cur.wait_stream(s)


# ---- raw launches (no autograd): the building blocks of Conv2dFunction and ResStageFunction -------------------

STEM7 = os.environ.get('HTD_STEM7', '1') != '0'       # 0: the stem on the generic kernels with an 8-channel image (A/B runs)


def _stem7_ok(x, weight, stride, padding, dilation, residual=None):
    """The ResNet stem in the form htd_conv2d_stem7_fwd takes: 7x7 / stride 2 / padding 3, 4 input channels (RGB + zero), 64
    output channels, fp32 with the split-bf16 arithmetic selected."""
    return (STEM7 and x.dtype == torch.float32 and tuple(weight.shape) == (64, 4, 7, 7) and x.size(1) == 4 and stride == 2 and
            padding == 3 and dilation == 1 and residual is None and capi.lib().htd_conv2d_set_math(-1) == 1)


# Activation planes (csrc/conv_x3.hip, conv_x3q_kernel): the 3x3 layer of a bottleneck writes the bf16 planes of its output next
# to the fp32 map and the 1x1 layer behind it reads both operands by LDS-DMA instead of splitting every element Co / 128 times
# in its K loop.  Measured per bottleneck (tools/bench_planes.py, profiles/r04_bench_planes.log, B = 4 @ 800x1344, with the
# residual epilogue): the plane-fed conv3 is 5.0 us faster in layer3 (59.4 -> 54.4) and 4.2 us in layer4, nothing in layer2 (its
# 128 -> 512 layer is bound by 276 MB of output + residual traffic, not by its loop), while the 3x3 producer pays 2.2 us
# (layer3), 4.4 us (layer2) and 5.1 us (layer4: 66 tiles, the planes come out of the K-range reduce pass) for writing them.
# So: where the consumer has >= ACT_PLANES_MIN_TILES = 8 column tiles of 128 AND the map has >= ACT_PLANES_MIN_ROWS = 8192
# pixels -- layer3 of the ResNets (23 blocks in R101).  HTD_ACT_PLANES=0 switches them off.
ACT_PLANES = os.environ.get('HTD_ACT_PLANES', '1') != '0'
ACT_PLANES_MIN_TILES = int(os.environ.get('HTD_ACT_PLANES_MIN_TILES', '8'))
ACT_PLANES_MIN_ROWS = int(os.environ.get('HTD_ACT_PLANES_MIN_ROWS', '8192'))


def _act_planes_buf(M, C, device):
    return torch.empty(capi.lib().htd_act_planes_bytes(M, C) // 4, device=device, dtype=torch.int32)


def act_planes(x):
    """x (B,C,H,W) fp32 channels_last, C % 16 == 0 -> its plane image (htd_act_planes), as a pass of its own."""
    x = x.contiguous(memory_format=CL)
    B, C, H, W = x.shape
    out = _act_planes_buf(B * H * W, C, x.device)
    capi.call('htd_act_planes', _P(x), _P(out), B * H * W, C, _S(), work=('byte', 10.0 * x.numel()))
    return out


def _planes_pay(w_consumer, transposed=False, rows=None):
    """Should the producer of the input of the 1x1 / stride-1 layer with this weight emit planes for it?  rows: pixels of the map."""
    if not ACT_PLANES or w_consumer.dim() != 4 or w_consumer.dtype != torch.float32:
        return False
    if H2_1X1 and capi.lib().htd_conv2d_set_h2(-1) == 1:
        return False            # the 1x1 consumer runs on H2 from the producer's carried maximum: nobody would read the planes
    if rows is not None and rows < ACT_PLANES_MIN_ROWS:
        return False
    Co, Ci, kh, kw = w_consumer.shape
    cred, cout = (Co, Ci) if transposed else (Ci, Co)
    return kh == 1 and kw == 1 and cred % 16 == 0 and (cout + 127) // 128 >= ACT_PLANES_MIN_TILES and \
        bool(capi.lib().htd_conv2d_x3p_supported(cred, cout, 1, 1, 1, 0, 1))


def _fwd_raw(x, weight, bias, residual, stride, padding, dilation, relu, res_up=False, x_planes=None, emit=None):
    """emit = True / False (not None): -> (y, planes), planes = the plane image of y when emit is True and this layer's kernel
    can write it, else None; x_planes: the planes of x (1x1 / stride-1 layers on conv_x3q_kernel)."""
    y = _fwd_raw_(x, weight, bias, residual, stride, padding, dilation, relu, res_up, x_planes, bool(emit))
    if emit is not None:
        return y if isinstance(y, tuple) else (y, None)
    return y


def _fwd_raw_(x, weight, bias, residual, stride, padding, dilation, relu, res_up, x_planes, emit):
    B, Ci, H, W = x.shape
    Co, Ci_w, kh, kw = weight.shape
    if Ci != Ci_w:
        raise ValueError(f'conv2d: input has {Ci} channels, weight expects {Ci_w}')
    Ho, Wo = _out_hw(H, W, kh, kw, stride, padding, dilation)
    y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=CL)
    if y.numel() == 0:              # empty batch (a stage without positive RoIs): nothing to launch
        return y
    flops = 2.0 * B * Ho * Wo * Co * kh * kw * Ci
    rh, rw = (residual.size(2), residual.size(3)) if (res_up and residual is not None) else (0, 0)
    if _stem7_ok(x, weight, stride, padding, dilation, residual):
        ws = torch.empty(capi.lib().htd_conv2d_stem7_workspace_bytes() // 4, device=x.device, dtype=torch.float32)
        capi.call('htd_conv2d_stem7_fwd', _P(x), _P(weight), _P(bias), _P(y), B, H, W, int(bool(relu)), _P(ws), _S(),
                  work=('flop', flops, 4.0 * (x.numel() + weight.numel() + y.numel())))
        return y
    if _x3p_ok(Ci, Co, kh, kw, stride, padding, dilation, x.dtype):
        nb = capi.lib().htd_conv2d_x3p_workspace_bytes(B * Ho * Wo, Co, Ci, kh, kw)
        ws = torch.empty(nb // 4, device=x.device, dtype=torch.float32) if nb > 0 else None
        use_xp = x_planes is not None and kh == 1 and stride == 1
        yp = _act_planes_buf(B * Ho * Wo, Co, x.device) if (emit and Co % 16 == 0) else None
        work = ('flop', flops, 4.0 * (x.numel() + weight.numel() + y.numel() + (y.numel() if residual is not None else 0)))
        h2 = _x3h_ok(Ci, Co, kh, kw, stride, padding, dilation, x.dtype)
        out_slot = _amax_slot(x.device) if h2 else None          # the epilogue leaves max |y| for the layer behind this one
        if h2:
            am = carried_amax(x)
            if am is None:
                _h2_trace('fwd', x, weight)
            if am is None and kh * kw * Co >= H2_ABSMAX_MIN_WORK and x.numel() >= H2_ABSMAX_MIN_ELEMS:
                am = absmax(x)
            if am is not None:
                capi.call('htd_conv2d_fwd_x3h', _P(x), _P(am), _P(x3_planes(weight, False, True)), _P(bias), _P(residual), rh, rw,
                          _P(y), _P(yp), _P(out_slot), B, H, W, Ci, Co, kh, kw, stride, padding,
                          int(bool(relu)), _P(ws), _S(), work=work)
                tag_amax(y, out_slot)
                _h2_guard(am, out_slot)
                return (y, yp) if emit else y
        # (algorithmic bytes stay those of the fp32 operands: the planes are this implementation's traffic, not the layer's)
        capi.call('htd_conv2d_fwd_x3q', _P(x), _P(x_planes) if use_xp else None, _P(x3_planes(weight, False, False)), _P(bias),
                  _P(residual), rh, rw, _P(y), _P(yp), _P(out_slot), B, H, W, Ci, Co, kh, kw, stride, padding, int(bool(relu)),
                  _P(ws), _S(), key='htd_conv2d_fwd_x3p', work=work)
        if out_slot is not None:
            tag_amax(y, out_slot)
        return (y, yp) if emit else y
    work = ('flop', flops, 4.0 * (x.numel() + weight.numel() + y.numel() + (y.numel() if residual is not None else 0)))
    if x.dtype == torch.float32 and rh == 0 and \
            capi.lib().htd_conv2d_x3h_strided_supported(Ci, Co, kh, kw, stride, padding, dilation):
        # the strided 3x3 of a stage's first block: nine taps on the H2 kernel's 1x1 loop when the input's maximum is known
        am = carried_amax(x)
        if am is None:
            _h2_trace('fwd', x, weight)
            if kh * kw * Co >= H2_ABSMAX_MIN_WORK and x.numel() >= H2_ABSMAX_MIN_ELEMS:
                am = absmax(x)
        if am is not None:
            nb = capi.lib().htd_conv2d_x3p_workspace_bytes(B * Ho * Wo, Co, Ci, kh * kw, 1)
            ws = torch.empty(nb // 4, device=x.device, dtype=torch.float32) if nb > 0 else None
            out_slot = _amax_slot(x.device)
            capi.call('htd_conv2d_fwd_x3h', _P(x), _P(am), _P(x3_planes(weight, False, True)), _P(bias), _P(residual), 0, 0, _P(y), None,
                      _P(out_slot), B, H, W, Ci, Co, kh, kw, stride, padding, int(bool(relu)), _P(ws), _S(), work=work)
            tag_amax(y, out_slot)
            _h2_guard(am, out_slot)
            return y
    if _igemm_emits(y, Co):
        # (a strided / dilated layer between layers that run on H2: its output's maximum rides along like theirs)
        slot = _amax_slot(x.device)
        capi.call('htd_conv2d_fwd_amax', _P(x), _P(weight), _P(bias), _P(residual), rh, rw, _P(y), _P(slot), B, H, W, Ci, Co, kh, kw,
                  stride, padding, dilation, int(bool(relu)), _P(_splitk_ws(B * Ho * Wo, Co, Ci, kh, kw, x.device)), _S(),
                  key='htd_conv2d_fwd', work=work)
        tag_amax(y, slot)
        return y
    capi.call('htd_conv2d_fwd', _P(x), _P(weight), _P(bias), _P(residual), rh, rw, _P(y), B, H, W, Ci, Co, kh, kw, stride,
              padding, dilation, int(bool(relu)), _P(_splitk_ws(B * Ho * Wo, Co, Ci, kh, kw, x.device)), _S(), work=work)
    return y


def _igemm_emits(y, C):
    """Does a conv_igemm_kernel launch leave the maximum of its output?  fp32 maps of at least 64 channels (what an H2 layer can
    read; the skinny heads' outputs go into losses) while H2 is on."""
    return y.dtype == torch.float32 and C >= 64 and emits('igemm')


def _colsum_raw(g, y=None, bias=None):
    """-> (gm, gbias): gbias = column sums of gm, gm = g * (y > 0) when y is given (else gm is g itself)."""
    B, Co, Ho, Wo = g.shape
    gm = torch.empty_like(g, memory_format=CL) if y is not None else g
    gb = grad_out(bias) if bias is not None else torch.empty(Co, device=g.device, dtype=g.dtype)
    if g.numel() == 0:
        return gm, gb.zero_()
    ws = torch.empty(2048 * Co, device=g.device, dtype=g.dtype)
    work = ('byte', 4.0 * B * Ho * Wo * Co * (3 if y is not None else 1))
    if _mask_emits(g):
        slot = _amax_slot(g.device)
        capi.call('htd_bias_grad_relu_mask_amax', _P(g), _P(y), _P(gm) if y is not None else None, _P(gb), B * Ho * Wo, Co,
                  _P(ws), _P(slot), _S(), key='htd_bias_grad_relu_mask', work=work)
        tag_amax(gm, slot)
    else:
        capi.call('htd_bias_grad_relu_mask', _P(g), _P(y), _P(gm) if y is not None else None, _P(gb), B * Ho * Wo, Co,
                  _P(ws), _S(), work=work)
    return gm, gb


# HTD_H2_EMIT_OFF=mask,igemm,roi,pool: producers that do NOT leave their output's maximum (A/B runs and bisection): the ReLU-mask /
# bias-gradient pass, conv_igemm_kernel's epilogues, RoIAlign and the global-context fusions, the stem's max pooling.
H2_EMIT_OFF = frozenset(v for v in os.environ.get('HTD_H2_EMIT_OFF', '').split(',') if v)


def emits(kind):
    return kind not in H2_EMIT_OFF and capi.lib().htd_conv2d_set_h2(-1) == 1


def _mask_emits(g):
    """The masked gradient goes into a data- and a weight-gradient launch: its maximum rides along when those can use it."""
    return g.dtype == torch.float32 and emits('mask')


def _mask_raw(g, y):
    """g * (y > 0): the ReLU backward alone (the bias gradient of the layer comes out of its wgrad launch)."""
    B, Co, Ho, Wo = g.shape
    gm = torch.empty_like(g, memory_format=CL)
    if g.numel() == 0:
        return gm
    if _mask_emits(g):
        slot = _amax_slot(g.device)
        capi.call('htd_bias_grad_relu_mask_amax', _P(g), _P(y), _P(gm), None, B * Ho * Wo, Co, None, _P(slot), _S(),
                  key='htd_bias_grad_relu_mask', work=('byte', 12.0 * B * Ho * Wo * Co))
        tag_amax(gm, slot)
    else:
        capi.call('htd_bias_grad_relu_mask', _P(g), _P(y), _P(gm), None, B * Ho * Wo, Co, None, _S(),
                  work=('byte', 12.0 * B * Ho * Wo * Co))
    return gm


def _dgrad_raw(g, weight, x_shape, stride, padding, dilation, mask_src=None, accum=None, wT=None, g_planes=None, emit=None):
    """Data gradient of conv2d(x, weight) for gy = g, optionally + accum and masked by (mask_src > 0).
    wT: the flipped / transposed image of `weight` when somebody made it already (htd_bn_fold_many_fwd).
    g_planes: the planes of g (1x1 layers); emit = True / False (not None): -> (gx, planes of gx or None)."""
    gx = _dgrad_raw_(g, weight, x_shape, stride, padding, dilation, mask_src, accum, wT, g_planes, bool(emit))
    if emit is not None:
        return gx if isinstance(gx, tuple) else (gx, None)
    return gx


def _dgrad_raw_(g, weight, x_shape, stride, padding, dilation, mask_src, accum, wT, g_planes, emit):
    B, Ci, H, W = x_shape
    Co, _, kh, kw = weight.shape
    Ho, Wo = g.shape[2], g.shape[3]
    if B == 0:
        return torch.empty((0, Ci, H, W), device=g.device, dtype=g.dtype, memory_format=CL)
    if stride == 1 and _x3p_ok(Co, Ci, kh, kw, 1, padding, dilation, g.dtype):
        gx = torch.empty((B, Ci, H, W), device=g.device, dtype=g.dtype, memory_format=CL)
        nb = capi.lib().htd_conv2d_x3p_workspace_bytes(B * H * W, Ci, Co, kh, kw)
        ws = torch.empty(nb // 4, device=g.device, dtype=torch.float32) if nb > 0 else None
        use_gp = g_planes is not None and kh == 1
        gxp = _act_planes_buf(B * H * W, Ci, g.device) if (emit and Ci % 16 == 0) else None
        work = ('flop', 2.0 * B * Ho * Wo * Co * kh * kw * Ci,
                4.0 * (g.numel() + weight.numel() + gx.numel() * (1 + (mask_src is not None) + (accum is not None))))
        h2 = _x3h_ok(Co, Ci, kh, kw, 1, padding, dilation, g.dtype)
        out_slot = _amax_slot(g.device) if h2 else None
        if h2:
            am = carried_amax(g)
            if am is None:
                _h2_trace('dgrad', g, weight)
            if am is None and kh * kw * Ci >= H2_ABSMAX_MIN_WORK and g.numel() >= H2_ABSMAX_MIN_ELEMS:
                am = absmax(g)
            if am is not None:
                capi.call('htd_conv2d_bwd_data_x3h', _P(g), _P(am), _P(x3_planes(weight, True, True)), _P(mask_src), _P(accum), _P(gx),
                          _P(gxp), _P(out_slot), B, H, W, Ci, Co, kh, kw, padding, _P(ws), _S(), work=work)
                tag_amax(gx, out_slot)
                _h2_guard(am, out_slot)
                return (gx, gxp) if emit else gx
        capi.call('htd_conv2d_bwd_data_x3q', _P(g), _P(g_planes) if use_gp else None, _P(x3_planes(weight, True, False)), _P(mask_src),
                  _P(accum), _P(gx), _P(gxp), _P(out_slot), B, H, W, Ci, Co, kh, kw, padding, _P(ws), _S(),
                  key='htd_conv2d_bwd_data_x3p', work=work)
        if out_slot is not None:
            tag_amax(gx, out_slot)
        return (gx, gxp) if emit else gx
    if stride != 1 and accum is None and g.dtype == torch.float32 and \
            capi.lib().htd_conv2d_bwd_data_x3h_strided_supported(Ci, Co, kh, kw, stride, padding, dilation):
        # the strided layers of a stage's first block: one H2 launch per parity class of gx's pixels when max |g| is known
        am = carried_amax(g)
        if am is None:
            _h2_trace('dgrad', g, weight)
            if kh * kw * Ci >= H2_ABSMAX_MIN_WORK and g.numel() >= H2_ABSMAX_MIN_ELEMS:
                am = absmax(g)
        if am is not None:
            gx = torch.empty((B, Ci, H, W), device=g.device, dtype=g.dtype, memory_format=CL)
            nb = capi.lib().htd_conv2d_bwd_data_x3h_strided_workspace_bytes(B, H, W, Ci, Co, kh, kw, stride, padding)
            ws = torch.empty(nb // 4, device=g.device, dtype=torch.float32) if nb > 0 else None
            out_slot = _amax_slot(g.device)
            capi.call('htd_conv2d_bwd_data_x3h_strided', _P(g), _P(am), _P(x3_planes(weight, True, True)), _P(mask_src), _P(gx),
                      _P(out_slot), B, H, W, Ci, Co, kh, kw, stride, padding, _P(ws), _S(), key='htd_conv2d_bwd_data_x3h',
                      work=('flop', 2.0 * B * Ho * Wo * Co * kh * kw * Ci,
                            4.0 * (g.numel() + weight.numel() + gx.numel() * (1 + (mask_src is not None)))))
            tag_amax(gx, out_slot)
            _h2_guard(am, out_slot)
            return gx
    gd, Cod = g, Co
    if Co % 8 != 0:      # skinny heads (RPN cls+reg Co=15, fc_cls 81, fc_reg 4): zero-pad the reduction channels
        Cod = Co + (-Co) % 8
        gd = torch.empty((B, Cod, Ho, Wo), device=g.device, dtype=g.dtype, memory_format=CL)
        if gd.numel():
            capi.call('htd_pad_channels', _P(g), _P(gd), B * Ho * Wo, Co, Cod, _S())
    if wT is None or Cod != Co:
        key = _flip_key(weight)                                               # one storage can be read as several shapes
        hit = _STEP_FLIPS.get(key) if Cod == Co else None
        wT = hit[1] if hit is not None else None
        if wT is None:
            wT = torch.empty(Ci * kh * kw * Cod, device=g.device, dtype=g.dtype)
            if Cod != Co:
                capi.call('htd_conv2d_flip_weights_padded', _P(weight), _P(wT), Co, Cod, kh, kw, Ci, _S())
            else:
                capi.call('htd_conv2d_flip_weights', _P(weight), _P(wT), Co, kh, kw, Ci, _S())
                _STEP_FLIPS[key] = (weight, wT)
    gx = torch.empty((B, Ci, H, W), device=g.device, dtype=g.dtype, memory_format=CL)
    work = ('flop', 2.0 * B * Ho * Wo * Co * kh * kw * Ci,
            4.0 * (g.numel() + weight.numel() + gx.numel() * (1 + (mask_src is not None) + (accum is not None))))
    if _igemm_emits(gx, Ci):
        slot = _amax_slot(g.device)
        capi.call('htd_conv2d_bwd_data_amax', _P(gd), _P(wT), _P(mask_src), _P(accum), _P(gx), _P(slot), B, H, W, Ci, Cod, kh, kw,
                  stride, padding, dilation, _P(_splitk_ws(B * H * W, Ci, Cod, kh, kw, g.device)), _S(),
                  key='htd_conv2d_bwd_data', work=work)
        tag_amax(gx, slot)
        return gx
    capi.call('htd_conv2d_bwd_data', _P(gd), _P(wT), _P(mask_src), _P(accum), _P(gx), B, H, W, Ci, Cod, kh, kw, stride,
              padding, dilation, _P(_splitk_ws(B * H * W, Ci, Cod, kh, kw, g.device)), _S(), work=work)
    return gx


SINK_ACCUMULATE = os.environ.get('HTD_SINK_ACC', '1') != '0'      # 0: shared parameters collect their gradient through autograd adds


H2_WGRAD = os.environ.get('HTD_H2_WGRAD', '1') != '0'          # 0: the weight gradients stay on the three-piece bf16 form


def _wgrad_amax(x, g, weight, stride, padding, dilation):
    """The two maxima of an H2 weight-gradient launch, or None when the layer stays on the bf16 form: both carried by the tensors
    (left by the epilogues that wrote them), or -- 3x3 layers only, where the arithmetic saves more than two passes cost --
    measured now.  Called on the stream that produced x and g, before the launch moves to the weight-gradient stream."""
    if not H2_WGRAD or x.dtype != torch.float32:
        return None
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight.shape
    if B * g.size(2) * g.size(3) == 0 or \
            not capi.lib().htd_conv2d_bwd_weight_h2_supported(B, H, W, Ci, Co, kh, kw, stride, padding, dilation):
        return None
    ax, ag = carried_amax(x), carried_amax(g)
    if ax is None:
        _h2_trace('wgrad x', x, weight)
    if ag is None:
        _h2_trace('wgrad g', g, weight)
    if (ax is None and (kh * kw * Co < H2_ABSMAX_MIN_WORK or x.numel() < H2_ABSMAX_MIN_ELEMS)) or \
            (ag is None and (kh * kw * Ci < H2_ABSMAX_MIN_WORK or g.numel() < H2_ABSMAX_MIN_ELEMS)):
        return None
    return (ax if ax is not None else absmax(x)), (ag if ag is not None else absmax(g))


def _wgrad_launch(x, g, weight, stride, padding, dilation, bias, amax=None):
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho, Wo = g.shape[2], g.shape[3]
    # A parameter shared by several layers (the RPN's convolutions on five pyramid levels, the stage-1 classifier that stage 2
    # reuses): its first gradient of the step was written into the flat-buffer slice; this one is ADDED there by the kernel
    # (htd_conv2d_bwd_weight_acc) and nothing is returned -- autograd has one defined contribution and no sums to launch.
    aw = grad_sink_again(weight) if SINK_ACCUMULATE else None
    ab = grad_sink_again(bias) if (aw is not None and torch.is_tensor(bias)) else None
    if aw is not None and (bias is None or ab is not None) and B * Ho * Wo > 0:
        nbytes = capi.lib().htd_conv2d_wgrad_workspace_bytes(B, H, W, Ci, Co, kh, kw, stride, padding, dilation)
        ws = torch.empty(nbytes // 4 + 1, device=g.device, dtype=g.dtype)
        work = ('flop', 2.0 * B * Ho * Wo * Co * kh * kw * Ci, 4.0 * (x.numel() + g.numel() + aw.numel()))
        if amax is not None:
            capi.call('htd_conv2d_bwd_weight_h2', _P(x), _P(g), _P(amax[0]), _P(amax[1]), _P(aw), _P(ab), B, H, W, Ci, Co, kh, kw,
                      stride, padding, dilation, 1, _P(ws), _S(), work=work)
        else:
            capi.call('htd_conv2d_bwd_weight_acc', _P(x), _P(g), _P(aw), _P(ab), B, H, W, Ci, Co, kh, kw, stride, padding,
                      dilation, _P(ws), _S(), key='htd_conv2d_bwd_weight', work=work)
        ent = _GRAD_SINK.get(weight.data_ptr())
        return None, None, not (ent is not None and ent[2] is _TEMP_USED)
    gw, sink_w = grad_out2(weight)
    gb, sink_b = (None, True)
    if bias is not None:            # True: bias gradient wanted, no parameter to look a sink up for
        gb, sink_b = grad_out2(bias) if torch.is_tensor(bias) else (torch.empty(Co, device=g.device, dtype=g.dtype), False)
    if B * Ho * Wo == 0:            # no pixels: the gradients are zeros
        gw.zero_()
        if gb is not None:
            gb.zero_()
        return gw, gb, sink_w and sink_b
    nbytes = capi.lib().htd_conv2d_wgrad_workspace_bytes(B, H, W, Ci, Co, kh, kw, stride, padding, dilation)
    ws = torch.empty(nbytes // 4 + 1, device=g.device, dtype=g.dtype)
    work = ('flop', 2.0 * B * Ho * Wo * Co * kh * kw * Ci, 4.0 * (x.numel() + g.numel() + gw.numel()))
    if amax is not None:
        capi.call('htd_conv2d_bwd_weight_h2', _P(x), _P(g), _P(amax[0]), _P(amax[1]), _P(gw), _P(gb), B, H, W, Ci, Co, kh, kw,
                  stride, padding, dilation, 0, _P(ws), _S(), work=work)
    else:
        capi.call('htd_conv2d_bwd_weight', _P(x), _P(g), _P(gw), _P(gb), B, H, W, Ci, Co, kh, kw, stride, padding,
                  dilation, _P(ws), _S(), work=work)
    return gw, gb, sink_w and sink_b


def _wgrad_raw(x, g, weight, stride, padding, dilation, bias=None, overlap=True):
    """-> (gw, gbias); gbias = column sums of g from the same launch when `bias` is given (the bias tensor, whose flat
    gradient slice is then written in place, or True), else None.  overlap=False keeps the launch on the main stream:
    for a caller that hands `g` itself on to autograd (record_stream guards against reuse of the storage, not against
    the engine's in-place `add_` into a gradient it owns while the side stream has not read it yet)."""
    amax = _wgrad_amax(x, g, weight, stride, padding, dilation)
    if not overlap or not OVERLAP_WGRAD or (capi.profiling() and not _OVERLAP_IN_PROFILE):
        if OVERLAP_WGRAD and _SIDE and SINK_ACCUMULATE and grad_sink_again(weight) is not None:
            # this launch ADDS into a slice whose earlier contributions may still be in the weight-gradient stream's queue
            torch.cuda.current_stream().wait_stream(side_stream(g.device))
        return _wgrad_launch(x, g, weight, stride, padding, dilation, bias, amax)[:2]
    main, side = torch.cuda.current_stream(), side_stream(g.device)
    side.wait_stream(main)
    x.record_stream(side)
    g.record_stream(side)
    with torch.cuda.stream(side):
        gw, gb, sinks = _wgrad_launch(x, g, weight, stride, padding, dilation, bias, amax)
    if not sinks and weight.data_ptr() not in _SIDE_CONSUMED:
        if gw is not None:
            gw.record_stream(main)
        if gb is not None:
            gb.record_stream(main)
        main.wait_stream(side)          # autograd consumes these gradients on the main stream
    return gw, gb


class Conv2dFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, padding, dilation, relu, res_up=False, chain=False):
        """chain=True: also returns an identity alias of x.  A later consumer of x that reads the alias instead hands its
        gradient to THIS node's backward, where it joins in the data-gradient epilogue (`accum`) -- one gradient map
        reaches x's producer and autograd has nothing to add (mmcv_ops.PyramidTaps does the same for the RoIAligns)."""
        _need_gpu(x, 'conv2d')
        src = x
        x = x.contiguous(memory_format=CL)
        weight = weight.contiguous(memory_format=CL)
        res = residual.contiguous(memory_format=CL) if residual is not None else None
        b = bias.contiguous() if bias is not None else None
        y = _fwd_raw(x, weight, b, res, stride, padding, dilation, relu, res_up)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.cfg = (stride, padding, dilation, bool(relu), bias is not None, residual is not None)
        ctx.res_up = tuple(res.shape) if (res_up and res is not None) else None
        ctx.bias_ref = b                                  # only its address is used (gradient sink lookup)
        ctx.chain = bool(chain)
        if chain:
            ctx.set_materialize_grads(False)              # an unused alias arrives as None, not as a map of zeros
            return y, src.view_as(src)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g, galias=None):
        x, weight, y = ctx.saved_tensors
        stride, padding, dilation, relu, has_bias, has_res = ctx.cfg
        if g is None:                                     # (chain) the convolution's own output was not used
            return (galias, ) + (None, ) * 9
        g = g.contiguous(memory_format=CL)
        need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
        gb = gw = None
        want_b = has_bias and need_b
        # the bias gradient is a by-product of the wgrad launch; without a weight gradient: stand-alone column sums
        if relu:
            if want_b and not need_w:
                g, gb = _colsum_raw(g, y, ctx.bias_ref)
            else:
                g = _mask_raw(g, y)
        elif want_b and not need_w:
            gb = _colsum_raw(g, None, ctx.bias_ref)[1]
        gx = None
        if need_x:
            fuse = galias is not None and stride == 1 and galias.dtype == g.dtype and tuple(galias.shape) == tuple(x.shape)
            gx = _dgrad_raw(g, weight, x.shape, stride, padding, dilation,
                            accum=galias.contiguous(memory_format=CL) if fuse else None)
            if galias is not None and not fuse:
                gx = gx + galias
        # a same-size residual gets `g` ITSELF as its gradient (below): autograd may then accumulate into that tensor in
        # place on the main stream, so this layer's weight gradient must not still be reading it on the side stream
        aliases_g = has_res and need_r and ctx.res_up is None
        if need_w:
            gw, gb = _wgrad_raw(x, g, weight, stride, padding, dilation, ctx.bias_ref if want_b else None,
                                overlap=not aliases_g)
        gr = None
        if has_res and need_r:
            gr = g
            if ctx.res_up is not None:      # residual came in through nearest up-sampling: sum the gradient back down
                rB, rC, rh, rw = ctx.res_up
                gr = torch.empty((rB, rC, rh, rw), device=g.device, dtype=g.dtype, memory_format=CL)
                capi.call('htd_upsample_nearest_bwd', _P(g), _P(gr), rB, g.size(2), g.size(3), rh, rw, rC, _S(),
                          work=('byte', 4.0 * (g.numel() + gr.numel())))
        return gx, gw, (gb if (has_bias and need_b) else None), gr, None, None, None, None, None, None


class ConvReluHeadFunction(Function):
    """y = conv1x1(relu(conv_kxk(x, w1) + b1), w2) + b2 as ONE autograd node (the RPN: rpn_head.py:24-33, a 3x3 convolution
    + ReLU feeding the 1x1 heads).  In backward the ReLU mask of the hidden map is applied by the head's data-gradient
    epilogue (mask_src) instead of a separate pass over the hidden gradient, and bias gradients come out of the wgrad
    launches.  chain=True: also returns an identity alias of x (see Conv2dFunction)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, padding, chain):
        _need_gpu(x, 'conv_relu_head')
        src = x
        x = x.contiguous(memory_format=CL)
        w1, w2 = w1.contiguous(memory_format=CL), w2.contiguous(memory_format=CL)
        b1 = b1.contiguous() if b1 is not None else None
        b2 = b2.contiguous() if b2 is not None else None
        h = _fwd_raw(x, w1, b1, None, 1, padding, 1, True)
        y = _fwd_raw(h, w2, b2, None, 1, 0, 1, False)
        ctx.save_for_backward(x, w1, w2, h)
        ctx.cfg = (int(padding), b1, b2, bool(chain))          # the biases: only their addresses are used (sink lookup)
        if chain:
            ctx.set_materialize_grads(False)
            return y, src.view_as(src)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, galias=None):
        x, w1, w2, h = ctx.saved_tensors
        padding, b1, b2, chain = ctx.cfg
        if gy is None:
            return (galias, ) + (None, ) * 6
        need_x, need_w1, need_b1, need_w2, need_b2 = ctx.needs_input_grad[:5]
        gy = gy.contiguous(memory_format=CL)
        gw1 = gb1 = gw2 = gb2 = None
        if need_w2:
            gw2, gb2 = _wgrad_raw(h, gy, w2, 1, 0, 1, b2 if (need_b2 and b2 is not None) else None)
        elif need_b2 and b2 is not None:
            gb2 = _colsum_raw(gy, None, b2)[1]
        gh = _dgrad_raw(gy, w2, h.shape, 1, 0, 1, mask_src=h)          # ReLU of the hidden map: in this epilogue
        if need_w1:
            gw1, gb1 = _wgrad_raw(x, gh, w1, 1, padding, 1, b1 if (need_b1 and b1 is not None) else None)
        elif need_b1 and b1 is not None:
            gb1 = _colsum_raw(gh, None, b1)[1]
        gx = None
        if need_x:
            fuse = galias is not None and galias.dtype == gh.dtype and tuple(galias.shape) == tuple(x.shape)
            gx = _dgrad_raw(gh, w1, x.shape, 1, padding, 1, accum=galias.contiguous(memory_format=CL) if fuse else None)
            if galias is not None and not fuse:
                gx = gx + galias
        elif galias is not None:
            gx = galias
        return gx, gw1, gb1, gw2, gb2, None, None


def conv_relu_head(x, w1, b1, w2, b2, padding=1, chain=False):
    return ConvReluHeadFunction.apply(x, w1, b1, w2, b2, int(padding), bool(chain))


# ---- grouped convolutions (ResNeXt conv2): slab-packed weights, csrc/gconv.hip ------------------------------------

def _gconv_pack(weight, groups, transpose):
    C, cg, kh, kw = weight.shape
    n = capi.lib().htd_gconv2d_packed_floats(C, groups, kh, kw)
    if n <= 0:
        raise ValueError(f'grouped conv2d: {C} channels in {groups} groups is unsupported')
    wp = torch.empty(n, device=weight.device, dtype=weight.dtype)
    capi.call('htd_gconv2d_pack_weights', _P(weight), _P(wp), C, groups, kh, kw, int(transpose), _S())
    return wp


def _gconv_fwd_raw(x, wp, bias, geom, relu, cols_shape=None):
    """x: image (B,C,H,W) channels_last, or the column buffer [M][taps][C] of a deformable conv (cols_shape=(B,H,W))."""
    C, groups, kh, kw, stride, padding, dilation = geom
    B, H, W = cols_shape if cols_shape is not None else (x.size(0), x.size(2), x.size(3))
    Ho, Wo = _out_hw(H, W, kh, kw, stride, padding, dilation)
    y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=CL)
    capi.call('htd_gconv2d_fwd', _P(x), _P(wp), _P(bias), _P(y), B, H, W, C, groups, kh, kw, stride, padding, dilation,
              int(bool(relu)), int(cols_shape is not None), _S(),
              work=('flop', 2.0 * B * Ho * Wo * C * kh * kw * (C // groups), 4.0 * (x.numel() + y.numel())))
    return y


def _gconv_dgrad_raw(g, wpT, geom, x_shape, cols=False):
    C, groups, kh, kw, stride, padding, dilation = geom
    B, _, H, W = x_shape
    M = g.size(0) * g.size(2) * g.size(3)
    gx = torch.empty((M, kh * kw, C), device=g.device, dtype=g.dtype) if cols else \
        torch.empty((B, C, H, W), device=g.device, dtype=g.dtype, memory_format=CL)
    capi.call('htd_gconv2d_bwd_data', _P(g), _P(wpT), _P(gx), B, H, W, C, groups, kh, kw, stride, padding, dilation,
              int(cols), _S(), work=('flop', 2.0 * M * C * kh * kw * (C // groups), 4.0 * (g.numel() + gx.numel())))
    return gx


def _gconv_wgrad_raw(x, g, weight, geom, x_shape, cols=False):
    C, groups, kh, kw, stride, padding, dilation = geom
    B, _, H, W = x_shape
    gw = grad_out(weight)
    nbytes = capi.lib().htd_gconv2d_wgrad_workspace_bytes(B, H, W, C, groups, kh, kw, stride, padding, dilation)
    ws = torch.empty(nbytes // 4 + 1, device=g.device, dtype=g.dtype)
    M = g.size(0) * g.size(2) * g.size(3)
    capi.call('htd_gconv2d_bwd_weight', _P(x), _P(g), _P(gw), B, H, W, C, groups, kh, kw, stride, padding, dilation,
              int(cols), _P(ws), _S(), work=('flop', 2.0 * M * C * kh * kw * (C // groups), 4.0 * (x.numel() + g.numel())))
    return gw


class GroupedConv2dFunction(Function):
    """y = act(conv2d(x, w, groups) + bias), in == out channels (the 3x3 of a ResNeXt bottleneck)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups, relu):
        _need_gpu(x, 'grouped conv2d')
        x = x.contiguous(memory_format=CL)
        weight = weight.contiguous(memory_format=CL)
        C, cg, kh, kw = weight.shape
        if x.size(1) != C or cg * groups != C:
            raise ValueError(f'grouped conv2d: input {tuple(x.shape)}, weight {tuple(weight.shape)}, groups {groups}')
        if x.dtype != torch.float32:
            raise TypeError('grouped conv2d runs in fp32')
        geom = (C, groups, kh, kw, stride, padding, dilation)
        b = bias.contiguous() if bias is not None else None
        y = _gconv_fwd_raw(x, _gconv_pack(weight, groups, False), b, geom, relu)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.geom, ctx.relu, ctx.bias_ref = geom, bool(relu), b
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        g = g.contiguous(memory_format=CL)
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        gb = None
        if ctx.bias_ref is not None and need_b:
            g, gb = _colsum_raw(g, y if ctx.relu else None, ctx.bias_ref)
        elif ctx.relu:
            g = _mask_raw(g, y)
        gx = _gconv_dgrad_raw(g, _gconv_pack(weight, ctx.geom[1], True), ctx.geom, x.shape) if need_x else None
        gw = _gconv_wgrad_raw(x, g, weight, ctx.geom, x.shape) if need_w else None
        return gx, gw, gb, None, None, None, None, None


def grouped_conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, relu=False):
    return GroupedConv2dFunction.apply(x, weight, bias, int(stride), int(padding), int(dilation), int(groups), bool(relu))


class ResStageFunction(Function):
    """A run of bottleneck blocks (one ResLayer, resnet.py:95-300 / res_layer.py:5-102) with frozen-BN-folded
    weights as ONE autograd node.  Forward is the same four fused convolutions per block as the per-layer path.
    The hand-written backward keeps the gradient chain inside the data-gradient epilogues:
      * the ReLU mask of each internal activation is applied by the dgrad that produces its gradient (mask_src) and
        the bias gradients come out of the wgrad launches: no separate pass over any gradient map;
      * the identity / downsample branch joins through `accum` in the conv1 dgrad epilogue (no separate add);
      * blocks after the first hand their predecessor a gradient already masked by the predecessor's output ReLU.
    args: x, strides (tuple), dilation, has_ds (tuple of bool), chain, then per block w1,b1,w2,b2,w3,b3[,wd,bd].
    chain=True: also returns an identity alias of x (see Conv2dFunction): a later consumer of x -- the FPN lateral of this
    pyramid level -- reads the alias, and its gradient joins in the first block's data-gradient epilogues (`accum`) instead
    of a full-map add by autograd (C3: 138 MB, C4: 69 MB per step)."""

    @staticmethod
    def forward(ctx, x, strides, dilation, has_ds, chain, *params):
        _need_gpu(x, 'res_stage')
        src = x
        x = x.contiguous(memory_format=CL)
        params = [t.contiguous(memory_format=CL) if t.dim() == 4 else t.contiguous() for t in params]
        saved, k = [], 0
        for stride, ds in zip(strides, has_ds):
            w1, b1, w2, b2, w3, b3 = params[k:k + 6]
            k += 6
            h1 = _fwd_raw(x, w1, b1, None, 1, 0, 1, True)
            # conv2 writes the bf16 planes of h2 for conv3 (1x1, Co = 4 x mid: every element of h2 would be split Co / 128 times)
            h2, h2p = _fwd_raw(h1, w2, b2, None, stride, dilation, dilation, True,
                               emit=_planes_pay(w3, rows=h1.size(0) * (h1.size(2) // stride) * (h1.size(3) // stride)))
            if ds:
                wd, bd = params[k:k + 2]
                k += 2
                idn = _fwd_raw(x, wd, bd, None, stride, 0, 1, False)
            else:
                idn = x
            out = _fwd_raw(h2, w3, b3, idn, 1, 0, 1, True, x_planes=h2p)
            saved += [x, h1, h2, out]
            x = out
        ctx.save_for_backward(*saved, *params)
        ctx.cfg = (strides, dilation, has_ds)
        ctx.flipped = [take_flipped(t) if t.dim() == 4 else None for t in params]      # same indexing as params
        if chain:
            ctx.set_materialize_grads(False)              # an unused alias arrives as None, not as a map of zeros
            return x, src.view_as(src)
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, g, galias=None):
        strides, dilation, has_ds = ctx.cfg
        if g is None:                                     # (chain) the stage's own output was not used
            return (galias, None, None, None, None) + (None, ) * (len(ctx.saved_tensors) - 4 * len(strides))
        nb = len(strides)
        saved, params = ctx.saved_tensors[:4 * nb], ctx.saved_tensors[4 * nb:]
        flipped = ctx.flipped
        offs, k = [], 0
        for ds in has_ds:
            offs.append(k)
            k += 8 if ds else 6
        need = ctx.needs_input_grad
        grads = [None] * len(params)
        g = g.contiguous(memory_format=CL)
        if galias is not None:
            galias = galias.contiguous(memory_format=CL)
        premasked = False
        for i in range(nb - 1, -1, -1):
            x, h1, h2, out = saved[4 * i:4 * i + 4]
            k, stride, ds = offs[i], strides[i], has_ds[i]
            w1, b1, w2, b2, w3, b3 = params[k:k + 6]
            pneed = need[5 + k:5 + k + (8 if ds else 6)]
            first = i == 0
            need_x = need[0] if first else True
            # bias gradients are by-products of the wgrad launches (column sums of the staged gy tiles); a conv whose
            # weight needs no gradient falls back to stand-alone column sums
            gm3 = g if premasked else _mask_raw(g, out)
            gb3 = None
            if pneed[4]:
                grads[k + 4], gb3 = _wgrad_raw(h2, gm3, w3, 1, 0, 1, True if (pneed[5] or (ds and pneed[7])) else None)
            elif pneed[5] or (ds and pneed[7]):
                gb3 = _colsum_raw(gm3)[1]
            grads[k + 5] = gb3 if pneed[5] else None
            gm2 = _dgrad_raw(gm3, w3, h2.shape, 1, 0, 1, mask_src=h2, wT=flipped[k + 4])
            if pneed[2]:
                grads[k + 2], gb2 = _wgrad_raw(h1, gm2, w2, stride, dilation, dilation, True if pneed[3] else None)
            else:
                gb2 = _colsum_raw(gm2)[1] if pneed[3] else None
            grads[k + 3] = gb2 if pneed[3] else None
            # (the planes of gm1 for conv1's data gradient, a 1x1 layer with 4 x mid or more output channels)
            gm1, gm1p = _dgrad_raw(gm2, w2, h1.shape, stride, dilation, dilation, mask_src=h1, wT=flipped[k + 2],
                                   emit=bool(need_x and _planes_pay(w1, True, rows=h1.size(0) * h1.size(2) * h1.size(3))))
            if pneed[0]:
                grads[k], gb1 = _wgrad_raw(x, gm1, w1, 1, 0, 1, True if pneed[1] else None)
            else:
                gb1 = _colsum_raw(gm1)[1] if pneed[1] else None
            grads[k + 1] = gb1 if pneed[1] else None
            acc = gm3
            if ds:
                wd = params[k + 6]
                if pneed[6]:
                    grads[k + 6] = _wgrad_raw(x, gm3, wd, stride, 0, 1)[0]
                grads[k + 7] = gb3 if pneed[7] else None
                # (first block: the chained consumer's gradient of x rides this epilogue, then conv1's below)
                acc = _dgrad_raw(gm3, wd, x.shape, stride, 0, 1, wT=flipped[k + 6],
                                 accum=galias if first else None) if need_x else None
            elif first and galias is not None and need_x:
                acc = gm3 + galias                        # (no downsample branch in the first block: not a ResNet stage)
            if need_x:
                g = _dgrad_raw(gm1, w1, x.shape, 1, 0, 1, mask_src=None if first else x, accum=acc, wT=flipped[k],
                               g_planes=gm1p)
                premasked = not first
            else:
                g = None
        if g is None and galias is not None and need[0]:
            g = galias
        return (g, None, None, None, None, *grads)


def conv2d_bf16(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, residual=None, res_up=False):
    """Forward-only bf16 convolution (fp32 accumulate, htd_conv2d_fwd_bf16): x (B,Ci,H,W) and weight (Co,Ci,kh,kw)
    bf16 channels_last, bias fp32, residual bf16 -> bf16.  Groundwork for the bf16 configurations; not yet wired into
    the detector (no gradient kernels)."""
    _need_gpu(x, 'conv2d_bf16')
    if x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
        raise TypeError('conv2d_bf16: x and weight must be bfloat16')
    x = x.contiguous(memory_format=CL)
    weight = weight.contiguous(memory_format=CL)
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho, Wo = _out_hw(H, W, kh, kw, stride, padding, dilation)
    y = torch.empty((B, Co, Ho, Wo), device=x.device, dtype=torch.bfloat16, memory_format=CL)
    b = bias.float().contiguous() if bias is not None else None
    res = residual.to(torch.bfloat16).contiguous(memory_format=CL) if residual is not None else None
    if res_up and res is not None:          # a coarser map added through nearest up-sampling (FPN top-down)
        if res.size(0) != B or res.size(1) != Co or res.size(2) > Ho or res.size(3) > Wo:
            raise ValueError('conv2d_bf16: up-sampled residual %s does not fit the output %s' % (tuple(res.shape), tuple(y.shape)))
        capi.call('htd_conv2d_fwd_bf16_up', _P(x), _P(weight), _P(b), _P(res), res.size(2), res.size(3), _P(y), B, H, W, Ci,
                  Co, kh, kw, int(stride), int(padding), int(dilation), int(bool(relu)), _S(), key='htd_conv2d_fwd_bf16',
                  work=('flop', 2.0 * B * Ho * Wo * Co * kh * kw * Ci, 2.0 * (x.numel() + weight.numel() + y.numel() + res.numel())))
        return y
    capi.call('htd_conv2d_fwd_bf16', _P(x), _P(weight), _P(b), _P(res), _P(y), B, H, W, Ci, Co, kh, kw, int(stride),
              int(padding), int(dilation), int(bool(relu)), _S(),
              work=('flop', 2.0 * B * Ho * Wo * Co * kh * kw * Ci,
                    2.0 * (x.numel() + weight.numel() + y.numel() * (2 if res is not None else 1))))
    return y


def conv2d_wgrad_bf16(x, gy, weight_shape, stride=1, padding=0, dilation=1, with_bias=False, weight=None, bias=None):
    """fp32 weight gradient of a convolution from bf16 activations x (B,Ci,H,W) and bf16 output gradient gy
    (htd_conv2d_bwd_weight_bf16) -> (Co,Ci,kh,kw) fp32 channels_last.  with_bias: -> (gw, gbias), the column sums of gy
    (fp32) from the same launch (htd_conv2d_bwd_weight_bf16_bias).  weight / bias (the fp32 master parameters, optional):
    their slices of the flat gradient buffer are written in place when they are registered as gradient sinks (grad_out2),
    as the fp32 path does -- no copy into the buffer afterwards."""
    _need_gpu(x, 'conv2d_wgrad_bf16')
    x = x.contiguous(memory_format=CL)
    gy = gy.contiguous(memory_format=CL)
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight_shape
    gw = None
    if weight is not None and weight.dtype == torch.float32 and tuple(weight.shape) == tuple(weight_shape):
        gw = grad_out2(weight)[0]
        if not (gw.dim() == 4 and gw.is_contiguous(memory_format=CL)):
            gw = None
    if gw is None:
        gw = torch.empty((Co, Ci, kh, kw), device=x.device, dtype=torch.float32, memory_format=CL)
    nbytes = capi.lib().htd_conv2d_wgrad_bf16_workspace_bytes(B, H, W, Ci, Co, kh, kw, stride, padding, dilation)
    ws = torch.empty(nbytes // 4 + 1, device=x.device, dtype=torch.float32)
    if with_bias:
        gb = grad_out2(bias)[0] if (bias is not None and bias.dtype == torch.float32 and bias.numel() == Co) else \
            torch.empty(Co, device=x.device, dtype=torch.float32)
        capi.call('htd_conv2d_bwd_weight_bf16_bias', _P(x), _P(gy), _P(gw), _P(gb), B, H, W, Ci, Co, kh, kw, int(stride),
                  int(padding), int(dilation), _P(ws), _S(), key='htd_conv2d_bwd_weight_bf16',
                  work=('flop', 2.0 * gy.numel() * kh * kw * Ci, 2.0 * (x.numel() + gy.numel()) + 4.0 * gw.numel()))
        return gw, gb
    capi.call('htd_conv2d_bwd_weight_bf16', _P(x), _P(gy), _P(gw), B, H, W, Ci, Co, kh, kw, int(stride), int(padding),
              int(dilation), _P(ws), _S(), work=('flop', 2.0 * gy.numel() * kh * kw * Ci, 2.0 * (x.numel() + gy.numel()) + 4.0 * gw.numel()))
    return gw


# ---- bf16 raw launches ------------------------------------------------------------------------------------------
BF16 = torch.bfloat16


def _prep_bf16(weight, want_wT=True):
    """fp32 (folded) weight (Co,Ci,kh,kw) channels_last -> (wb, wT): bf16 forward operand and the transposed,
    tap-flipped data-gradient operand, one launch."""
    weight = weight.contiguous(memory_format=CL)
    Co, Ci, kh, kw = weight.shape
    wb = torch.empty((Co, Ci, kh, kw), device=weight.device, dtype=BF16, memory_format=CL)
    wT = torch.empty((Ci, Co, kh, kw), device=weight.device, dtype=BF16, memory_format=CL) if want_wT else None
    capi.call('htd_weights_prep_bf16', _P(weight), _P(wb), _P(wT), Co, kh, kw, Ci, _S())
    return wb, wT


def _prep_bf16_many(items):
    """[(fp32 weight (Co,Ci,kh,kw), want_wT)] -> [(wb, wT or None)]: the bf16 operands of many layers in ONE launch
    (htd_weights_prep_bf16_many) instead of one ~6-microsecond launch per layer (R101: 104 per step)."""
    import numpy as np
    if not items:
        return []
    ws = [w.contiguous(memory_format=CL) for w, _ in items]
    dev = ws[0].device
    total = sum(w.numel() * (2 if want else 1) for w, (_, want) in zip(ws, items))
    flat = torch.empty(total, device=dev, dtype=BF16)
    desc = np.zeros((len(ws), 6), dtype=np.int64)
    out, off, tile0 = [], 0, 0
    for i, (w, (_, want)) in enumerate(zip(ws, items)):
        Co, Ci, kh, kw = w.shape
        wb = flat[off:off + w.numel()].view(Co, kh, kw, Ci).permute(0, 3, 1, 2)           # KRSC memory = channels_last
        off += w.numel()
        wT = None
        if want:
            wT = flat[off:off + w.numel()].view(Ci, kh, kw, Co).permute(0, 3, 1, 2)
            off += w.numel()
        desc[i, 0], desc[i, 1], desc[i, 2] = w.data_ptr(), wb.data_ptr(), wT.data_ptr() if wT is not None else 0
        desc[i, 3] = Co | ((kh * kw) << 32)
        desc[i, 4] = Ci
        desc[i, 5] = tile0
        tile0 += kh * kw * ((Co + 31) // 32) * ((Ci + 31) // 32)
        out.append((wb, wT))
    table = capi.upload_table(desc, dev)
    capi.call('htd_weights_prep_bf16_many', _P(table), len(ws), tile0, _S())
    return out


def _dgrad_bf16_raw(g, wT, kh, pad, dil, mask_src=None, accum=None):
    """gx = (conv(g, wT) + accum) * (mask_src > 0) for a stride-1 layer with padding `pad`; all maps bf16."""
    B, Co, H, W = g.shape
    Ci = wT.size(0)
    p = dil * (kh - 1) - pad
    Ho, Wo = _out_hw(H, W, kh, kh, 1, p, dil)
    gx = torch.empty((B, Ci, Ho, Wo), device=g.device, dtype=BF16, memory_format=CL)
    capi.call('htd_conv2d_dgrad_bf16', _P(g), _P(wT), _P(mask_src), _P(accum), _P(gx), B, H, W, Co, Ci, kh, kh, p, dil,
              _S(), work=('flop', 2.0 * B * Ho * Wo * Ci * kh * kh * Co,
                          2.0 * (g.numel() + wT.numel() + gx.numel() * (1 + (mask_src is not None) + (accum is not None)))))
    return gx


def _colsum_bf16_raw(g, out=None):
    """fp32 column sums of the bf16 gradient map g (B,C,H,W) channels_last -> (C,)."""
    B, C, H, W = g.shape
    rows = B * H * W
    out = torch.empty(C, device=g.device, dtype=torch.float32) if out is None else out
    nbytes = capi.lib().htd_colsum_bf16_workspace_bytes(rows, C)
    ws = torch.empty(nbytes // 4 + 1, device=g.device, dtype=torch.float32)
    capi.call('htd_colsum_bf16', _P(g), _P(out), rows, C, _P(ws), _S())
    return out


def _dgrad_bf16_any(g, wT, w32, x_shape, kh, stride, pad, dil, mask_src=None, accum=None):
    """Data gradient in bf16; strided layers (three per backbone) fall back to the fp32 kernel."""
    if stride == 1:
        return _dgrad_bf16_raw(g, wT, kh, pad, dil, mask_src, accum)
    gx = _dgrad_raw(g.float().contiguous(memory_format=CL), w32, x_shape, stride, pad, dil,
                    mask_src=mask_src.float() if mask_src is not None else None,
                    accum=accum.float() if accum is not None else None)
    return gx.to(BF16)


class Conv2dBf16Function(Function):
    """Mixed-precision convolution for the bf16 configurations: bf16 activations, fp32 master weight / bias.
    forward  y = act(conv(x, bf16(w)) + b [+ residual])           htd_conv2d_fwd_bf16
    backward gx = conv(gy, flip(bf16(w)))  (stride 1; strided layers fall back to the fp32 kernels)
             gw = fp32 weight gradient from bf16 x, gy           htd_conv2d_bwd_weight_bf16
    Gradients w.r.t. weight and bias come out in fp32, ready for the flat fp32 gradient buffer."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, padding, dilation, relu, res_up=False):
        need_wT = ctx.needs_input_grad[0] and stride == 1
        wb, wT = _prep_bf16(weight, need_wT)
        y = conv2d_bf16(x, wb, bias, stride, padding, dilation, relu, residual, res_up)
        ctx.save_for_backward(x, wT if need_wT else wb, y if relu else None)
        ctx.cfg = (stride, padding, dilation, bias is not None, residual is not None, tuple(weight.shape), need_wT)
        ctx.res_up = tuple(residual.shape) if (res_up and residual is not None) else None
        ctx.master = (weight, bias)                 # the parameters themselves: their gradient sinks are looked up in backward
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, wsaved, y = ctx.saved_tensors
        stride, padding, dilation, has_bias, has_res, wshape, have_wT = ctx.cfg
        w_master, b_master = ctx.master
        g = g.contiguous(memory_format=CL)
        if y is not None:
            g = torch.ops.aten.threshold_backward(g, y, 0)          # ReLU backward: one launch
        need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
        gx = gw = gb = None
        kh = wshape[2]
        if need_x:
            if have_wT:
                gx = _dgrad_bf16_raw(g, wsaved, kh, padding, dilation)
            else:
                gx = _dgrad_raw(g.float().contiguous(memory_format=CL), wsaved.float().contiguous(memory_format=CL),
                                x.shape, stride, padding, dilation).to(torch.bfloat16)
        if need_w and has_bias and need_b:
            gw, gb = conv2d_wgrad_bf16(x, g, wshape, stride, padding, dilation, with_bias=True, weight=w_master,
                                       bias=b_master)      # one launch for both
        else:
            if need_w:
                gw = conv2d_wgrad_bf16(x, g, wshape, stride, padding, dilation, weight=w_master)
            if has_bias and need_b:
                gb = _colsum_bf16_raw(g) if g.size(1) % 4 == 0 else \
                    torch.sum(g.permute(0, 2, 3, 1).reshape(-1, g.size(1)), dim=0, dtype=torch.float32)
        gr = None
        if has_res and need_r:
            gr = g
            if ctx.res_up is not None:      # the residual came in through nearest up-sampling: sum the gradient back down
                rB, rC, rh, rw = ctx.res_up
                gr = torch.empty((rB, rC, rh, rw), device=g.device, dtype=g.dtype, memory_format=CL)
                capi.call('htd_upsample_nearest_bwd_bf16', _P(g), _P(gr), rB, g.size(2), g.size(3), rh, rw, rC, _S(),
                          work=('byte', 2.0 * (g.numel() + gr.numel())))
        return gx, gw, gb, gr, None, None, None, None, None


class ResStageBf16Function(Function):
    """ResStageFunction for bf16 activations (fp32 BN-folded master parameters): one autograd node per ResLayer.
    Per layer and step one launch prepares both bf16 weight operands; in backward the ReLU masks and the identity-branch
    sum live in the data-gradient epilogues (htd_conv2d_dgrad_bf16), bias gradients are one column-sum launch each.
    Strided data gradients (the first block of layers 2-4) fall back to the fp32 kernel."""

    @staticmethod
    def forward(ctx, x, strides, dilation, has_ds, *params):
        _need_gpu(x, 'res_stage_bf16')
        x = x.contiguous(memory_format=CL)
        params = [t.contiguous(memory_format=CL) if t.dim() == 4 else t.contiguous() for t in params]
        saved, wTs, k = [], [], 0
        bwd = any(ctx.needs_input_grad)           # inference: forward operands only
        todo, kk = [], 0                          # the bf16 operands of every layer of the stage: one launch
        for stride, ds in zip(strides, has_ds):
            todo += [(params[kk], bwd), (params[kk + 2], bwd and stride == 1), (params[kk + 4], bwd)]
            kk += 6
            if ds:
                todo.append((params[kk], bwd and stride == 1))
                kk += 2
        prepped = iter(_prep_bf16_many(todo))
        for stride, ds in zip(strides, has_ds):
            w1, b1, w2, b2, w3, b3 = params[k:k + 6]
            k += 6
            (wb1, wT1), (wb2, wT2), (wb3, wT3) = next(prepped), next(prepped), next(prepped)
            h1 = conv2d_bf16(x, wb1, b1, 1, 0, 1, True)
            h2 = conv2d_bf16(h1, wb2, b2, stride, dilation, dilation, True)
            wTd = None
            if ds:
                wd, bd = params[k:k + 2]
                k += 2
                wbd, wTd = next(prepped)
                idn = conv2d_bf16(x, wbd, bd, stride, 0, 1, False)
            else:
                idn = x
            out = conv2d_bf16(h2, wb3, b3, 1, 0, 1, True, idn)
            saved += [x, h1, h2, out]
            wTs += [wT1, wT2, wT3, wTd]
            x = out
        ctx.save_for_backward(*saved, *params)
        ctx.wTs = wTs                      # bf16 operands of this step (never exposed to autograd)
        ctx.cfg = (strides, dilation, has_ds)
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        strides, dilation, has_ds = ctx.cfg
        nb = len(strides)
        saved, params = ctx.saved_tensors[:4 * nb], ctx.saved_tensors[4 * nb:]
        wTs = ctx.wTs
        offs, k = [], 0
        for ds in has_ds:
            offs.append(k)
            k += 8 if ds else 6
        need = ctx.needs_input_grad
        grads = [None] * len(params)
        g = g.to(BF16).contiguous(memory_format=CL)
        premasked = False
        for i in range(nb - 1, -1, -1):
            x, h1, h2, out = saved[4 * i:4 * i + 4]
            wT1, wT2, wT3, wTd = wTs[4 * i:4 * i + 4]
            k, stride, ds = offs[i], strides[i], has_ds[i]
            w1, b1, w2, b2, w3, b3 = params[k:k + 6]
            pneed = need[4 + k:4 + k + (8 if ds else 6)]
            first = i == 0
            need_x = need[0] if first else True
            gm3 = g if premasked else torch.ops.aten.threshold_backward(g, out, 0)
            # bias (= BN beta) gradients are by-products of the wgrad launches; stand-alone column sums only without a wgrad
            want_b3 = pneed[5] or (ds and pneed[7])
            gb3 = None
            if pneed[4] and want_b3 and w3.size(0) % 8 == 0:
                grads[k + 4], gb3 = conv2d_wgrad_bf16(h2, gm3, w3.shape, 1, 0, 1, with_bias=True)
            else:
                if pneed[4]:
                    grads[k + 4] = conv2d_wgrad_bf16(h2, gm3, w3.shape, 1, 0, 1)
                gb3 = _colsum_bf16_raw(gm3) if want_b3 else None
            grads[k + 5] = gb3 if pneed[5] else None
            gm2 = _dgrad_bf16_raw(gm3, wT3, 1, 0, 1, mask_src=h2)
            if pneed[2] and pneed[3]:
                grads[k + 2], grads[k + 3] = conv2d_wgrad_bf16(h1, gm2, w2.shape, stride, dilation, dilation, with_bias=True)
            else:
                if pneed[2]:
                    grads[k + 2] = conv2d_wgrad_bf16(h1, gm2, w2.shape, stride, dilation, dilation)
                grads[k + 3] = _colsum_bf16_raw(gm2) if pneed[3] else None
            gm1 = _dgrad_bf16_any(gm2, wT2, w2, h1.shape, 3, stride, dilation, dilation, mask_src=h1)
            if pneed[0] and pneed[1]:
                grads[k], grads[k + 1] = conv2d_wgrad_bf16(x, gm1, w1.shape, 1, 0, 1, with_bias=True)
            else:
                if pneed[0]:
                    grads[k] = conv2d_wgrad_bf16(x, gm1, w1.shape, 1, 0, 1)
                grads[k + 1] = _colsum_bf16_raw(gm1) if pneed[1] else None
            acc = gm3
            if ds:
                wd = params[k + 6]
                if pneed[6]:
                    grads[k + 6] = conv2d_wgrad_bf16(x, gm3, wd.shape, stride, 0, 1)
                grads[k + 7] = gb3 if pneed[7] else None
                acc = _dgrad_bf16_any(gm3, wTd, wd, x.shape, 1, stride, 0, 1) if need_x else None
            if need_x:
                g = _dgrad_bf16_raw(gm1, wT1, 1, 0, 1, mask_src=None if first else x, accum=acc)
                premasked = not first
            else:
                g = None
        ctx.wTs = None
        return (g, None, None, None, *grads)


def conv2d_bf16_autograd(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, residual=None, residual_up=False):
    """Differentiable bf16 convolution with fp32 master parameters (see Conv2dBf16Function).  residual_up: the residual is a
    coarser bf16 map, added through nearest-neighbour up-sampling to the output size."""
    return Conv2dBf16Function.apply(x, weight, bias, residual, int(stride), int(padding), int(dilation), bool(relu),
                                    bool(residual_up))


def _pad_channels(x, weight, mult=8):
    """Zero-pad the channel dimension of (x, weight) to a multiple of `mult` (3-channel stem input)."""
    Ci = x.size(1)
    pad = (-Ci) % mult
    if pad == 0:
        return x, weight
    x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, pad))
    weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, pad))
    return x.contiguous(memory_format=CL), weight.contiguous(memory_format=CL)


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, residual=None, residual_up=False, chain=False):
    """y = act(conv2d(x, w) + bias + residual); x (B,Ci,H,W) channels_last, weight (Co,Ci,kh,kw) channels_last.
    residual_up: residual is a coarser map, added through nearest-neighbour up-sampling to the output size.
    chain: -> (y, alias of x), see Conv2dFunction.forward."""
    if isinstance(stride, (tuple, list)):
        stride, padding, dilation = stride[0], padding[0], dilation[0]
    if x.size(1) % 8 != 0:
        assert not chain
        # the ResNet stem on an image nobody differentiates: RGB + ONE zero channel, htd_conv2d_stem7_fwd / a 4-channel weight
        # gradient (half the reduction-side work of the 8-channel form); everything else: channels padded to a multiple of 8
        stem = (STEM7 and x.size(1) == 3 and tuple(weight.shape) == (64, 3, 7, 7) and int(stride) == 2 and int(padding) == 3 and
                int(dilation) == 1 and residual is None and not x.requires_grad and x.dtype == torch.float32 and x.is_cuda and
                capi.lib().htd_conv2d_set_math(-1) == 1)
        x, weight = _pad_channels(x, weight, 4 if stem else 8)
    return Conv2dFunction.apply(x, weight, bias, residual, int(stride), int(padding), int(dilation), relu,
                                bool(residual_up), bool(chain))


def linear(x, weight, bias=None, relu=False):
    """y = act(x @ weight.T + bias); x (M,K), weight (N,K): the 1x1 convolution over M 'pixels'."""
    M, K = x.shape
    N = weight.size(0)
    if M == 0:
        return x.new_zeros(0, N) + (0 * weight.sum())
    if K % 8 != 0:
        pad = (-K) % 8
        x = torch.nn.functional.pad(x, (0, pad))
        weight = torch.nn.functional.pad(weight, (0, pad))
        K += pad
    x4 = x.contiguous().view(M, 1, 1, K).permute(0, 3, 1, 2)          # (M, K, 1, 1) channels_last view
    w4 = weight.contiguous().view(N, 1, 1, K).permute(0, 3, 1, 2)
    if x.dtype == torch.bfloat16 and K % 32 == 0 and N % 8 == 0:     # bf16 activations: mixed-precision kernels
        y = Conv2dBf16Function.apply(x4, w4, bias, None, 1, 0, 1, relu)
    else:
        y = Conv2dFunction.apply(x4.float() if x.dtype != torch.float32 else x4, w4, bias, None, 1, 0, 1, relu)
    return y.permute(0, 2, 3, 1).reshape(M, N)                         # (M, N, 1, 1) channels_last -> (M, N)


BGEMM_BWD_LIMITS = os.environ.get('HTD_BGEMM_BWD_LIMITS', '1') != '0'      # 0: full padded gradient products (A/B runs)


class BatchedGemmNT(Function):
    """c[g] = a[g] @ b[g]^T on the MFMA kernel (htd_bgemm_nt).  a (G,M,K), b (G,N,K) -> (G,M,N).
    Backward is two more NT products on transposed copies: ga = gc @ b, gb = gc^T @ a."""

    @staticmethod
    def _run(a, b, counts=None, limit=0):
        G, M, K = a.shape
        N = b.size(1)
        c = torch.empty(G, M, N, device=a.device, dtype=a.dtype)
        if counts is not None and limit and M % 128 == 0:
            # zero-padded groups: tiles and reduction ranges beyond the group's size are skipped on the device
            capi.call('htd_bgemm_nt_counts', _P(a), _P(b), _P(c), G, M, N, K, _P(counts), int(limit), _S(),
                      work=('flop', 2.0 * G * M * N * K))
        else:
            capi.call('htd_bgemm_nt', _P(a), _P(b), _P(c), G, M, N, K, _S(), work=('flop', 2.0 * G * M * N * K))
        return c

    @staticmethod
    def forward(ctx, a, b, counts=None, limit=0):
        _need_gpu(a, 'bgemm_nt')
        ctx.gram = a is b                       # a @ a^T (PGraph's similarity): one gradient product instead of two
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        ctx.counts, ctx.limit = counts, int(limit)
        return BatchedGemmNT._run(a, b, counts, limit)

    @staticmethod
    @once_differentiable
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        gc = gc.contiguous()
        ga = gb = None
        # the gradient products skip the padding as the forward product did: an axis that was limited to the group's size keeps
        # its limit in the role it plays now (bits: 1 rows of the first operand, 2 rows of the second, 4 reduction)
        counts, lim = (ctx.counts, ctx.limit) if BGEMM_BWD_LIMITS else (None, 0)
        lim_a = (lim & 1) | ((lim & 4) >> 1) | ((lim & 2) << 1)      # ga = gc @ b: rows M, columns K (was the reduction), reduction N
        lim_b = ((lim & 2) >> 1) | ((lim & 4) >> 1) | ((lim & 1) << 2)     # gb = gc^T @ a: rows N, columns K, reduction M
        if ctx.gram and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # d(a a^T): gc @ a + gc^T @ a = (gc + gc^T) @ a, handed back through the first argument
            return BatchedGemmNT._run(gc + gc.transpose(1, 2), a.transpose(1, 2).contiguous(), counts, lim_a), None, None, None
        if ctx.needs_input_grad[0]:
            ga = BatchedGemmNT._run(gc, b.transpose(1, 2).contiguous(), counts, lim_a)   # (G,M,N) x (G,K,N)^T
        if ctx.needs_input_grad[1]:
            gb = BatchedGemmNT._run(gc.transpose(1, 2).contiguous(), a.transpose(1, 2).contiguous(), counts, lim_b)   # (G,N,M) x (G,K,M)^T
        return ga, gb, None, None


def bgemm_nt(a, b, counts=None, limit=0):
    """Batched a @ b^T; every dimension that becomes a row count must be a multiple of 128 when G > 1 and every
    reduction length a multiple of 8 (PGraph pads its groups accordingly).  counts (G,) int64 device tensor + limit bits
    (1 rows of a, 2 rows of b, 4 reduction): group g has counts[g] real entries along those axes, zeros beyond -- the padding
    is then neither read nor multiplied (htd_bgemm_nt_counts)."""
    return BatchedGemmNT.apply(a, b, counts, int(limit))


def roofline_report(prof, peak_tflops, peak_gbs, peak_bf16_tflops=2500.0):
    """The `roofline` object of bench.py for the dominant hand-written kernel of the timed region:
    achieved = algorithmic work of its launches / their summed device time (live HIP-event timing)."""
    # forward and data-gradient calls run the same GPU kernel (conv_igemm_kernel): one class, as rocprofv3 sees it
    prof = dict(prof)
    parts = [prof.pop(k) for k in ('htd_conv2d_fwd', 'htd_conv2d_bwd_data') if k in prof]
    if parts:
        prof['conv_igemm_kernel (htd_conv2d_fwd + htd_conv2d_bwd_data)'] = (
            sum(p[0] for p in parts), sum(p[1] for p in parts), 'flop', sum(p[3] for p in parts), sum(p[4] for p in parts))
    parts = [prof.pop(k) for k in ('htd_conv2d_fwd_x3p', 'htd_conv2d_bwd_data_x3p') if k in prof]
    if parts:                                        # the pre-split-weights / halo kernel of csrc/conv_x3.hip
        prof['conv_x3p_kernel (htd_conv2d_fwd_x3p + htd_conv2d_bwd_data_x3p)'] = (
            sum(p[0] for p in parts), sum(p[1] for p in parts), 'flop', sum(p[3] for p in parts), sum(p[4] for p in parts))
    parts = [prof.pop(k) for k in ('htd_conv2d_fwd_bf16', 'htd_conv2d_dgrad_bf16') if k in prof]
    if parts:                                        # same for the bf16 kernel
        prof['conv_bf16_kernel (htd_conv2d_fwd_bf16 + htd_conv2d_dgrad_bf16)'] = (
            sum(p[0] for p in parts), sum(p[1] for p in parts), 'flop', sum(p[3] for p in parts), sum(p[4] for p in parts))
    parts = [prof.pop(k) for k in ('htd_conv2d_fwd_x3h', 'htd_conv2d_bwd_data_x3h') if k in prof]
    if parts:                                        # the same kernel on the H2 arithmetic (three fp16 products per fp32 product)
        prof['conv_x3p_kernel, H2 form (htd_conv2d_fwd_x3h + htd_conv2d_bwd_data_x3h)'] = (
            sum(p[0] for p in parts), sum(p[1] for p in parts), 'flop', sum(p[3] for p in parts), sum(p[4] for p in parts))
    if 'htd_conv2d_bwd_weight' in prof:
        prof['conv_wgrad_kernel + splitk_reduce_kernel (htd_conv2d_bwd_weight)'] = prof.pop('htd_conv2d_bwd_weight')
    if 'htd_conv2d_bwd_weight_h2' in prof:
        prof['conv_wgrad_x3d / x3hd kernels, H2 form + splitk_reduce_kernel (htd_conv2d_bwd_weight_h2)'] = prof.pop('htd_conv2d_bwd_weight_h2')
    best = None
    for name, (calls, ms, kind, work, nbytes) in prof.items():
        if kind is None or ms <= 0:
            continue
        if best is None or ms > best[2]:
            best = (name, calls, ms, kind, work, nbytes)
    if best is None:
        return None
    name, calls, ms, kind, work, nbytes = best
    if kind == 'flop':
        if 'bf16' in name:                       # bf16 matrix-core peak for the bf16 kernels
            peak_tflops = peak_bf16_tflops
        elif 'H2 form' in name:                  # three fp16 products per fp32 product: the 16-bit matrix peak / 3
            peak_tflops = peak_bf16_tflops / 3.0
        achieved = work / (ms * 1e-3) / 1e12
        out = dict(kernel=name, bound='mfma', achieved=round(achieved, 3), peak=peak_tflops, unit='TFLOP/s',
                   frac=round(achieved / peak_tflops, 4), traffic=None, launches=calls,
                   avg_launch_ms=round(ms / calls, 4), algorithmic_bytes=int(nbytes / calls))
        if 'bf16' in name and nbytes > 0:
            # the bf16 configurations sit nearer the HBM roof than the matrix roof ("report both fractions"): price the
            # class against the roof it is closer to, keep the other fraction beside it
            gbs = nbytes / (ms * 1e-3) / 1e9
            out['hbm'] = dict(achieved=round(gbs, 1), peak=peak_gbs, unit='GB/s', frac=round(gbs / peak_gbs, 4),
                              what='algorithmic bytes (bf16 operands and result once, fp32 weight gradients) / device time')
            if gbs / peak_gbs > achieved / peak_tflops:
                out.update(bound='hbm', mfma=dict(achieved=out['achieved'], peak=peak_tflops, unit='TFLOP/s', frac=out['frac']),
                           achieved=round(gbs, 1), peak=peak_gbs, unit='GB/s', frac=round(gbs / peak_gbs, 4))
        return out
    achieved = work / (ms * 1e-3) / 1e9
    return dict(kernel=name, bound='hbm', achieved=round(achieved, 2), peak=peak_gbs, unit='GB/s',
                frac=round(achieved / peak_gbs, 4), traffic=None, launches=calls, avg_launch_ms=round(ms / calls, 4))
