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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
lv = map_roi_levels(rois, 4)
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
print('RoIs per level', torch.bincount(lv, minlength=4).tolist())
fd = [f.clone().requires_grad_() for f in feats]
out = M.roi_align_levels(fd, rois, lv, 7, scales)
go = torch.randn_like(out)
map_bytes = sum(f.numel() for f in feats) * 4
roi_bytes = n * 49 * 256 * 4


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


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
