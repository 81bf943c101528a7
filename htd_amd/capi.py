"""ctypes binding of libhtd_amd.so -- the C-ABI drop-in boundary (include/htd_amd.h).

The prototypes are read from the header itself, so the Python argtypes can never drift
from the declared ABI.  There is NO fallback: if the shared library is missing or a
symbol is absent, importing/using the ops raises.  (`python -m htd_amd.csrc.build` or
`__graft_entry__.build()` compiles it for gfx950.)
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_PKG), 'include', 'htd_amd.h')
LIB_PATH = os.environ.get('HTD_AMD_LIB') or os.path.join(_PKG, 'libhtd_amd.so')      # HTD_AMD_LIB: an experiment build, for A/B runs

_PROTO = re.compile(r'^\s*(const char \*|int64_t|int)\s*(htd_\w+)\s*\(([^;]*?)\)\s*;', re.M | re.S)


def _ctype(decl):
    decl = decl.strip()
    if '*' in decl:
        return ctypes.c_void_p
    base = decl.rsplit(' ', 1)[0].strip() if ' ' in decl else decl
    base = base.replace('const', '').strip()
    return {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float,
            'uint8_t': ctypes.c_uint8}[base]


def declared_functions(header=HEADER):
    """[(name, restype, [argtypes])] for every prototype in include/htd_amd.h."""
    text = re.sub(r'/\*.*?\*/', '', open(header).read(), flags=re.S)
    out = []
    for ret, name, args in _PROTO.findall(text):
        args = ' '.join(args.split())
        argtypes = [] if args in ('', 'void') else [_ctype(a) for a in args.split(',')]
        restype = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'const char *': ctypes.c_char_p}[ret.strip()]
        out.append((name, restype, argtypes))
    return out


class HtdError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library with typed prototypes; raises if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: the HTD MI355X ops have no fallback path. '
                'Build it with `python -m htd_amd.csrc.build` (needs hipcc, targets gfx950).')
        L = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in declared_functions():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        want = int(re.search(r'#define\s+HTD_ABI_VERSION\s+(\d+)', open(HEADER).read()).group(1))
        got = L.htd_abi_version()
        if got != want:                     # a stale .so next to a newer header (checked once after dlopen)
            raise ImportError(f'{LIB_PATH}: ABI version {got}, include/htd_amd.h declares {want}: rebuild the library')
        _lib = L
        from . import tuning
        tuning.load(L)                      # tile table measured on MI355X (htd_amd/tuning/conv_tiles_gfx950.json)
    return _lib


def check(status, what=''):
    if status != 0:
        msg = lib().htd_last_error().decode()
        if status == 1:
            raise ValueError(f'{what}: {msg}')
        raise HtdError(f'{what}: {msg}')


# ---------------------------------------------------------------------------------- live kernel timing
# bench.py brackets every C-ABI call with device events recorded on the stream the kernel is launched on
# (torch's current stream, which is the stream handed to the library), so per-kernel-class durations are
# measured inside the timed region itself.  `work` = (kind, amount): algorithmic FLOPs ('flop') or bytes
# ('byte') of that call, summed per entry point for the roofline line.
_PROFILE = None
_PAUSED = False


_DETAIL = False


def profiling():
    """True while bench.py's per-call device-event timing is active and not paused (calls must then stay on one stream)."""
    return _PROFILE is not None and not _PAUSED


_ONLY = None
_PAUSED = False


def profile_pause(paused=True):
    """Suspend / resume event timing inside a profile_begin..profile_end region (bench.py times every n-th step)."""
    global _PAUSED
    _PAUSED = bool(paused)


def profile_begin(detail=False, only=None):
    """detail=True keys the records by entry point AND integer arguments (layer shapes); only = entry points to time
    (None = all).  Two events per call cost ~2.5 us of queue time each: timing all ~1800 calls of a train step
    lengthens it by 6 %, so the default benchmark times the matrix-core entry points only."""
    global _PROFILE, _DETAIL, _ONLY
    _PROFILE = {}
    _DETAIL = detail
    _ONLY = None if only is None else frozenset(only)
    profile_pause(False)


def profile_end():
    """-> {name: (calls, total_ms, work_kind, total_work, total_algorithmic_bytes)}; synchronises the device."""
    global _PROFILE
    import torch
    prof, _PROFILE = _PROFILE, None
    if prof is None:
        return {}
    torch.cuda.synchronize()
    out = {}
    for name, rec in prof.items():
        ms = sum(a.elapsed_time(b) for a, b in rec['events'])
        out[name] = (len(rec['events']), ms, rec['kind'], rec['work'], rec['bytes'])
    return out


def call(name, *args, work=None, key=None):
    """key: the entry point this call is accounted under in the live timing (a variant of an entry point that belongs to the
    same kernel class, e.g. htd_conv2d_fwd_x3q under htd_conv2d_fwd_x3p); default: its own name."""
    fn = getattr(lib(), name)
    key = key or name
    if _PROFILE is None or _PAUSED or (_ONLY is not None and key not in _ONLY):
        check(fn(*args), name)
        return
    import torch
    if _DETAIL:
        key = key + '(' + ','.join(str(a) for a in args if isinstance(a, int) and not isinstance(a, bool)) + ')'
        # which pointer operands are present (residual / mask / accum select epilogue variants): 1 = given, 0 = NULL
        key += '[' + ''.join('0' if a is None else '1' for a in args if a is None or isinstance(a, ctypes.c_void_p)) + ']'
    rec = _PROFILE.setdefault(key, dict(events=[], kind=None, work=0.0, bytes=0.0))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    status = fn(*args)
    b.record()
    rec['events'].append((a, b))
    if work is not None:
        rec['kind'] = work[0]
        rec['work'] += float(work[1])
        if len(work) > 2:
            rec['bytes'] += float(work[2])              # algorithmic HBM bytes of the call (operands once)
    check(status, name)


_RAW_STREAM = None


def current_stream_ptr():
    """The HIP stream torch is issuing on for the current device, as a pointer argument (a few hundred calls per step: the raw
    query, not a torch.cuda.Stream object per call)."""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch
        raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        if raw is not None:
            _RAW_STREAM = lambda: raw(torch.cuda.current_device())
        else:
            _RAW_STREAM = lambda: torch.cuda.current_stream().cuda_stream
    return ctypes.c_void_p(_RAW_STREAM())


# Descriptor tables of the multi-tensor launches (BN folds, plane images, weight preparation: one row of pointers and sizes per
# tensor).  The caching allocator hands out the same addresses step after step, so a step's tables are byte for byte the
# previous step's: content-addressed, each distinct table crosses PCIe once instead of once per step (a pinned allocation, a
# copy and ~40 us of host time each; ~20 per train step).
_TABLES = {}
_TABLE_CACHE = os.environ.get('HTD_TABLE_CACHE', '1') != '0'


def upload_table(desc, dev):
    """numpy int64 table -> device tensor with the same content (cached by content)."""
    import torch
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), desc.tobytes())
    t = _TABLES.get(key) if _TABLE_CACHE else None
    if t is None:
        if len(_TABLES) >= 512:
            _TABLES.clear()
        t = torch.from_numpy(desc.reshape(-1).copy()).pin_memory().to(dev, non_blocking=True)
        _TABLES[key] = t
    return t


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
