#!/usr/bin/env python3
"""RoIAlign forward / backward (scatter with atomics vs gather form) on the P2..P5 pyramid of B images at 800x1344:
time per call and HBM-side algorithmic rate.  usage: bench_roi_align.py [n_rois] [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd import capi
from htd_amd import mmcv_ops as M

ARGS = [a for a in sys.argv[1:] if not a.startswith('--')]
n = int(ARGS[0]) if len(ARGS) > 0 else 2048
B = int(ARGS[1]) if len(ARGS) > 1 else 4
dev = torch.device('cuda:0')
CL = torch.channels_last
g = torch.Generator().manual_seed(0)
feats = [torch.randn(B, 256, 800 // s, 1344 // s, device=dev).contiguous(memory_format=CL) for s in (4, 8, 16, 32)]
# log-uniform box sizes 16..600 px: the level mix of a trained detector's proposals
size = torch.exp(torch.rand(n, generator=g) * (6.4 - 2.8) + 2.8)
ar = torch.exp(torch.rand(n, generator=g) - 0.5)
w, h = size * ar, size / ar
cx, cy = torch.rand(n, generator=g) * 1333, torch.rand(n, generator=g) * 800
rois = torch.stack([torch.sort(torch.randint(0, B, (n, ), generator=g).float())[0], (cx - w / 2).clamp(0, 1333), (cy - h / 2).clamp(0, 800),
                    (cx + w / 2).clamp(0, 1333), (cy + h / 2).clamp(0, 800)], 1).to(dev)
from htd_amd.detector.roi_extractors import map_roi_levels


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


if '--ba' in sys.argv:
    # BA (AdptRoIExtractor): n positive RoIs, jittered copies of a few gt boxes per image (bench.py trained_like_proposals), pooled
    # from EVERY level: per-level launches against the one-launch forms.  HTD_ROI_BWD_SPLIT=1/2/4 sets the wavefronts per strip.
    gts = torch.rand(B, 5, 4, generator=g)
    img = torch.sort(torch.randint(0, B, (n, ), generator=g))[0]
    pick = gts[img, torch.randint(0, 5, (n, ), generator=g)]
    cx, cy = pick[:, 0] * 1333, pick[:, 1] * 800
    w, h = 30 + pick[:, 2] * 600, 30 + pick[:, 3] * 400
    jit = (torch.rand(n, 4, generator=g) - 0.5) * 0.12 * torch.stack([w, h, w, h], 1)
    rois = torch.stack([img.float(), (cx - w / 2 + jit[:, 0]).clamp(0, 1333), (cy - h / 2 + jit[:, 1]).clamp(0, 800),
                        (cx + w / 2 + jit[:, 2]).clamp(0, 1333), (cy + h / 2 + jit[:, 3]).clamp(0, 800)], 1).to(dev)
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    fd = [f.clone().requires_grad_() for f in feats]
    gos = [torch.randn(n, 256, 7, 7, device=dev).contiguous(memory_format=CL) for _ in range(4)]
    t_lvl = timed(lambda: [M.roi_align(feats[i], rois, 7, scales[i], 0, 'avg', True) for i in range(4)])
    t_all = timed(lambda: M.roi_align_all_levels(feats, rois, 7, scales))

    def bwd(one):
        for f in fd:
            f.grad = None
        outs = M.roi_align_all_levels(fd, rois, 7, scales) if one else [M.roi_align(fd[i], rois, 7, scales[i], 0, 'avg', True) for i in range(4)]
        torch.autograd.backward(outs, gos)
    tb_lvl = timed(lambda: bwd(False)) - t_lvl
    tb_all = timed(lambda: bwd(True)) - t_all
    print(f'BA {n} RoIs on 4 levels (split {os.environ.get("HTD_ROI_BWD_SPLIT", "default")}): forward per level {t_lvl * 1e6:8.1f} us, one launch {t_all * 1e6:8.1f} us; '
          f'backward per level {tb_lvl * 1e6:8.1f} us, one launch {tb_all * 1e6:8.1f} us')
    sys.exit(0)
lv = map_roi_levels(rois, 4)
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
print('RoIs per level', torch.bincount(lv, minlength=4).tolist())
fd = [f.clone().requires_grad_() for f in feats]
out = M.roi_align_levels(fd, rois, lv, 7, scales)
go = torch.randn_like(out)
map_bytes = sum(f.numel() for f in feats) * 4
roi_bytes = n * 49 * 256 * 4


t = timed(lambda: M.roi_align_levels(feats, rois, lv, 7, scales))
print(f'forward  {n} RoIs: {t * 1e6:8.1f} us   (writes {roi_bytes / 1e6:.0f} MB)')
for mode in ('scatter', 'gather/level', 'gather'):
    M.ROI_BWD = mode.split('/')[0]
    M.ROI_BWD_ONE_LAUNCH = mode == 'gather'         # 'gather': all levels in one launch

    def bwd():
        for f in fd:
            f.grad = None
        o = M.roi_align_levels(fd, rois, lv, 7, scales)
        o.backward(go)
    tb = timed(bwd) - t
    nbytes = map_bytes + roi_bytes + (map_bytes if mode == 'scatter' else 0)        # scatter: memset + read-modify-write
    print(f'backward {mode:14s}: {tb * 1e6:8.1f} us   {map_bytes / 1e6:.0f} MB of gradient maps + {roi_bytes / 1e6:.0f} MB read = '
          f'{(map_bytes + roi_bytes) / tb / 1e12:.2f} TB/s algorithmic')
