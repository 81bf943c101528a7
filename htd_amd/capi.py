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
LIB_PATH = os.path.join(_PKG, 'libhtd_amd.so')

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
        _lib = L
    return _lib


def check(status, what=''):
    if status != 0:
        msg = lib().htd_last_error().decode()
        if status == 1:
            raise ValueError(f'{what}: {msg}')
        raise HtdError(f'{what}: {msg}')


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
