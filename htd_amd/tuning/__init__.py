"""Tuned launch parameters measured on MI355X (gfx950), loaded into libhtd_amd.so when the library is first used.

`conv_tiles_gfx950.json` -- tile configuration of conv_igemm_kernel per convolution problem, measured INSIDE the HTD
train / inference steps by tools/tune_conv_tiles.py (see include/htd_amd.h: htd_conv2d_tile_table_set).  What the
reference gets from cuDNN's algorithm search (`cudnn_benchmark`), as a table that ships with the package: no search
at run time, same choice on every run.  HTD_CONV_TABLE=0 disables it (heuristic score only).
"""
import json
import os

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_tiles_gfx950.json')
X3P_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv_x3p_tiles_gfx950.json')      # conv_x3p_kernel (csrc/conv_x3.hip)


def read_table(path=TABLE):
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return {tuple(int(v) for v in k.split(',')): int(c) for k, c in json.load(f).get('entries', {}).items()}


def write_table(entries, meta=None, path=TABLE):
    data = dict(meta=meta or {}, entries={','.join(str(v) for v in k): int(c) for k, c in sorted(entries.items())})
    with open(path, 'w') as f:
        json.dump(data, f, indent=0, sort_keys=True)
        f.write('\n')


def load(lib, path=TABLE):
    """Push the table into the loaded library.  -> number of entries."""
    if os.environ.get('HTD_CONV_TABLE', '1') == '0':
        return 0
    n = 0
    for (M, Co, Ci, taps, epi), cfg in read_table(path).items():
        if lib.htd_conv2d_tile_table_set(M, Co, Ci, taps, epi, cfg) == 0:
            n += 1
    if path == TABLE:
        for (M, Co, Ci, taps, epi), cfg in read_table(X3P_TABLE).items():
            if lib.htd_conv2d_x3p_tile_table_set(M, Co, Ci, taps, epi, cfg) == 0:
                n += 1
    return n
