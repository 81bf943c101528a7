#!/usr/bin/env python3
"""Measure the tile configuration of conv_igemm_kernel per convolution problem INSIDE the HTD step and write the table
htd_amd/tuning/conv_tiles_gfx950.json (loaded by htd_amd/capi.py; include/htd_amd.h: htd_conv2d_tile_table_set).

For every configuration id the whole step (train: fwd + bwd + SGD; --infer: simple_test) runs with that tile forced
on every htd_conv2d_fwd / htd_conv2d_bwd_data launch (HTD_CONV_TUNE / HTD_CONV_FORCE_TILE), timed per call with device
events; a problem gets a table entry when its best tile beats the launcher's own score-based choice by more than
--margin.  Entries of earlier runs (other models / batch sizes) are kept.

    python tools/tune_conv_tiles.py [--depth 101] [--dcn] [--trained-like] [--infer --batch 64] [--steps 3]
"""
import argparse
import os
import re
import sys

os.environ['HTD_CONV_TUNE'] = '1'
os.environ['HTD_X3P_TUNE'] = '1'
os.environ['HTD_CONV_TABLE'] = '0'          # start from the heuristic; the table is applied by hand below
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

KEY = re.compile(r'^(htd_conv2d_fwd|htd_conv2d_bwd_data|htd_conv2d_fwd_x3p|htd_conv2d_bwd_data_x3p|htd_conv2d_fwd_x3h|htd_conv2d_bwd_data_x3h)'
                 r'\(([-\d,]+)\)\[([01]+)\]$')


def signature(name, ints, mask):
    """(M, Co, Ci, taps, epi) as conv_fwd.hip::launch_conv keys the launch, or None when the table does not apply."""
    # (calls of htd_conv2d_fwd_x3q / _bwd_data_x3q are accounted under the x3p names, dense.py.  Pointer slots:
    #  x3q  x, xplanes, wplanes, bias, residual, y, yplanes, amax_out, ws, stream  /  gy, gyplanes, wplanesT, mask_src, accum, gx, ...
    #  x3h  x, amax, wplanes, bias, residual, y, yplanes, amax_out, h2_flag, ws, stream  /  gy, amax, wplanesT, mask_src, accum, gx, ...
    #  x3p  x, wplanes, bias, residual, y, ws, stream  /  gy, wplanesT, mask_src, accum, gx, ws, stream)
    q = len(mask) >= 9
    h2 = 4 if name.endswith('x3h') else 0    # table key bit 2: the launch runs on the H2 arithmetic
    if name in ('htd_conv2d_fwd_x3p', 'htd_conv2d_fwd_x3h'):         # conv_x3.hip::launch_x3p
        res_h, res_w, B, H, W, Ci, Co, kh, kw, stride, pad, relu = ints
        Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        taps = kh * kw + (100 if (kh == 3 and stride == 2) else 0)          # conv_x3.hip::launch_x3p: tap-list launches key with + 100
        return (B * Ho * Wo, Co, Ci, taps, int(mask[4 if q else 3] == '1') | h2)
    if name in ('htd_conv2d_bwd_data_x3p', 'htd_conv2d_bwd_data_x3h'):
        if len(ints) != 8:
            # htd_conv2d_bwd_data_x3h_strided (gy, amax, wplanesT, mask_src, gx, amax_out, ws, stream): one launch per parity class
            # of gx's pixels, timed together -- ('strided', key of class 0, key of class 1, ...), all classes get the call's best tile
            B, H, W, Ci, Co, kh, kw, stride, pad = ints
            epi = (int(mask[3] == '1') << 1) | 4
            keys = []
            for c in range(stride * stride):
                ph, pw = c // stride, c % stride
                ny = sum(1 for ky in range(kh) if (ph + pad - ky) % stride == 0)
                nx = sum(1 for kx in range(kw) if (pw + pad - kx) % stride == 0)
                Hc, Wc = (H - ph + stride - 1) // stride if ph < H else 0, (W - pw + stride - 1) // stride if pw < W else 0
                if ny * nx and Hc * Wc:
                    keys.append((B * Hc * Wc, Ci, Co, ny * nx + 100, epi))
            return ('strided', ) + tuple(keys)
        B, H, W, Ci, Co, kh, kw, pad = ints
        return (B * H * W, Ci, Co, kh * kw, int(mask[4 if q else 3] == '1') | (int(mask[3 if q else 2] == '1') << 1) | h2)
    if name == 'htd_conv2d_fwd':
        res_h, res_w, B, H, W, Ci, Co, kh, kw, stride, pad, dil, relu = ints
        Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
        return (B * Ho * Wo, Co, Ci, kh * kw, int(mask[3] == '1'))
    B, H, W, Ci, Co, kh, kw, stride, pad, dil = ints
    if stride != 1:
        return None                          # tap-table sub-problems: scored, not tabled
    return (B * H * W, Ci, Co, kh * kw, int(mask[3] == '1') | (int(mask[2] == '1') << 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--dcn', action='store_true')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--infer', action='store_true')
    ap.add_argument('--proposals', type=int, default=512)
    ap.add_argument('--trained-like', action='store_true')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--margin', type=float, default=0.02)
    ap.add_argument('--kernel', default='igemm', choices=('igemm', 'x3p'),
                    help='igemm: conv_igemm_kernel (conv_tiles_gfx950.json); x3p: conv_x3p_kernel (conv_x3p_tiles_gfx950.json)')
    ap.add_argument('--dry', action='store_true', help='measure and print, do not write the table')
    ap.add_argument('--table', default=None, help='table file to merge into (default: the in-tree one)')
    args = ap.parse_args()
    import bench
    from htd_amd import capi, tuning
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.runner import Trainer, synthetic_batch
    dev = torch.device('cuda:0')
    L = capi.lib()
    torch.manual_seed(0)
    data = synthetic_batch(args.batch, 800, 1344, 1333, device=dev, seed=0)
    if args.infer:
        cfg = htd_config(args.depth, dcn=args.dcn, soft_nms=False)
        cfg.test_cfg.rpn.update(nms_post=args.proposals, max_num=args.proposals)
        model = build_htd_detector(cfg=cfg).to(dev).eval()

        def step():
            with torch.no_grad():
                model.simple_test(data['img'], data['img_metas'])
    else:
        model = build_htd_detector(args.depth, dcn=args.dcn).to(dev).train()
        trainer = Trainer(model, lr=1e-4)
        if args.trained_like:
            bench.trained_like_proposals(model, data, args.batch)

        def step():
            trainer.train_step(data)

    x3p = args.kernel == 'x3p'
    only = ('htd_conv2d_fwd_x3p', 'htd_conv2d_bwd_data_x3p', 'htd_conv2d_fwd_x3h', 'htd_conv2d_bwd_data_x3h') if x3p else \
        ('htd_conv2d_fwd', 'htd_conv2d_bwd_data')
    ids = (0, 1, 2, 3) if x3p else (0, 2, 3, 4, 5)
    force = 'HTD_X3P_FORCE_TILE' if x3p else 'HTD_CONV_FORCE_TILE'
    query = L.htd_conv2d_x3p_tile_query if x3p else L.htd_conv2d_tile_query
    default_table = tuning.X3P_TABLE if x3p else tuning.TABLE
    L.htd_conv2d_x3p_tile_table_clear()
    times = {}                               # cfg -> {signature: [total_ms, calls]}
    for cfg_id in (-1, ) + ids:
        os.environ[force] = str(cfg_id)
        step()
        capi.profile_begin(detail=True, only=only)
        for _ in range(args.steps):
            step()
        per = {}
        for key, (calls, ms, *_rest) in capi.profile_end().items():
            m = KEY.match(key)
            if not m or m.group(1) not in only:
                continue
            sig = signature(m.group(1), [int(v) for v in m.group(2).split(',')], m.group(3))
            if sig is None:
                continue
            rec = per.setdefault(sig, [0.0, 0])
            rec[0] += ms
            rec[1] += calls
        times[cfg_id] = per
        print(f'# cfg {cfg_id:2d}: {sum(v[0] for v in per.values()) / args.steps:8.2f} ms of tabled conv time per step',
              flush=True)
    os.environ[force] = '-1'
    path = args.table or default_table
    table = tuning.read_table(path if os.path.exists(path) else default_table)
    gained = 0.0
    print(f'{"M":>8s} {"Co":>5s} {"Ci":>5s} taps epi | {"auto":>8s} ' + ' '.join(f'{"cfg" + str(c):>8s}' for c in ids) +
          ' | pick   us/call')
    for sig in sorted(times[-1], key=lambda s_: -times[-1][s_][0]):
        auto_ms, calls = times[-1][sig]
        if sig[0] == 'strided':             # the parity classes of one strided data gradient: one measurement, one pick for all of them
            cand = {c: times[c][sig][0] for c in ids if sig in times[c]}
            best = min(cand, key=cand.get)
            pick = best if cand[best] < auto_ms * (1.0 - args.margin) else None
            for key in sig[1:]:
                if pick is not None and query(*key) != pick:
                    table[key] = pick
                elif pick is None and key in table:
                    del table[key]
            if pick is not None:
                gained += (auto_ms - cand[best]) / args.steps
            k0 = sig[1]
            print(f'{k0[0]:8d} {k0[1]:5d} {k0[2]:5d} {k0[3]:4d} {k0[4]:3d} | {auto_ms / calls * 1e3:8.1f} ' +
                  ' '.join(f'{cand[c] / calls * 1e3:8.1f}' if c in cand else f'{"-":>8s}' for c in ids) +
                  f' | {("cfg" + str(pick)) if pick is not None else "auto":7s} x{calls // args.steps} (strided data gradient, {len(sig) - 1} classes)')
            continue
        auto_cfg = query(*sig)
        cand = {}
        for c in ids:
            if not x3p and sig[1] <= 64 and c in (3, 5):
                continue                     # bn = 128 on a narrow output: never
            if sig in times[c]:
                cand[c] = times[c][sig][0]
        best = min(cand, key=cand.get)
        pick = best if cand[best] < auto_ms * (1.0 - args.margin) and best != auto_cfg else None
        if pick is not None:
            table[sig] = pick
            gained += (auto_ms - cand[best]) / args.steps
        elif sig in table:
            del table[sig]                   # the score already picks (about) the best: no entry needed
        print(f'{sig[0]:8d} {sig[1]:5d} {sig[2]:5d} {sig[3]:4d} {sig[4]:3d} | {auto_ms / calls * 1e3:8.1f} ' +
              ' '.join(f'{cand[c] / calls * 1e3:8.1f}' if c in cand else f'{"-":>8s}' for c in ids) +
              f' | {("cfg" + str(pick)) if pick is not None else "auto=" + str(auto_cfg):7s} x{calls // args.steps}')
    print(f'# expected gain over the score-based choice: {gained:.2f} ms per step; table now has {len(table)} entries')
    if not args.dry:
        tuning.write_table(table, meta=dict(device='MI355X (gfx950)', tool='tools/tune_conv_tiles.py' + (' --kernel x3p' if x3p else ''),
                                            note='tile id per (M,Co,Ci,taps,epi); see include/htd_amd.h'), path=path)


if __name__ == '__main__':
    main()
