#!/usr/bin/env python3
"""PGraph (htd_bbox_head.py:195-219) alone at the inference shape of BASELINE configs[4]: B = 64 images x 512 proposals,
4 levels, fc 1024, semantic embedding 256.  Forward time of htd_amd.detector.pgraph.pgraph_refine and its share of FLOP.
usage: bench_pgraph.py [B] [proposals]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htd_amd.detector.pgraph import pgraph_refine  # noqa: E402
from htd_amd.detector.roi_extractors import map_roi_levels  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
N = B * P
xy = torch.rand(N, 2, generator=g) * torch.tensor([1200., 700.])
wh = torch.exp(torch.randn(N, 2, generator=g) * 0.8 + 4.2).clamp(max=600.)
rois = torch.cat([torch.arange(B).repeat_interleave(P)[:, None].float(), xy, xy + wh], 1).to(dev)
lvls = map_roi_levels(rois, 4)
x = torch.randn(N, 1024, device=dev)
sam = torch.randn(N, 256, device=dev) * 0.1
layers = torch.nn.ModuleList([torch.nn.Linear(1024, 1024) for _ in range(4)]).to(dev)
counts = torch.bincount((rois[:, 0].long() * 4 + lvls), minlength=B * 4)
print('RoIs per (image, level) group: max %d, mean %.1f; padded group size %d' %
      (int(counts.max()), float(counts.float().mean()), (P + 127) // 128 * 128))
with torch.no_grad():
    for _ in range(3):
        pgraph_refine(x, sam, rois, lvls, layers, rois_per_img=[P] * B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        pgraph_refine(x, sam, rois, lvls, layers, rois_per_img=[P] * B)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
npad = (P + 127) // 128 * 128
G = B * 4
flop_padded = 2.0 * G * npad * npad * (1024 * 2 + 256) + 2.0 * G * npad * 1024 * 1024
flop_real = 2.0 * float((counts.double() ** 2).sum()) * (1024 * 2 + 256) + 2.0 * N * 1024 * 1024
print('pgraph_refine forward: %.3f ms for %d RoIs (%d groups); padded-batch FLOP %.1f G (%.1f TF/s), FLOP of the real '
      'groups %.1f G' % (ms, N, G, flop_padded / 1e9, flop_padded / ms / 1e9, flop_real / 1e9))
