"""Time the fused data-pipeline kernel (htd_image_batch_pipeline) on a COCO-shaped batch already resident in HBM.
usage: python tools/bench_pipeline.py [--batch 4] [--iters 50]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from htd_amd.pipelines import DeferredImage, DeviceBatchStager, rescale_size  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--iters', type=int, default=50)
ap.add_argument('--scale', type=int, nargs=2, default=(1333, 800))
args = ap.parse_args()
rs = np.random.RandomState(0)
shapes = [(480, 640), (427, 640), (640, 480), (375, 500), (500, 333), (612, 612)]
imgs = []
for b in range(args.batch):
    h, w = shapes[b % len(shapes)]
    d = DeferredImage(rs.randint(0, 256, (h, w, 3)).astype(np.uint8))
    (nw, nh), _ = rescale_size((w, h), tuple(args.scale))
    d.out_hw, d.flip = (nh, nw), ('horizontal' if b % 2 else None)
    d.norm = (np.float32([123.675, 116.28, 103.53]), np.float32([58.395, 57.12, 57.375]), True)
    d.pad_hw = (-(-nh // 32) * 32, -(-nw // 32) * 32)
    imgs.append(d)
st = DeviceBatchStager('cuda:0')
dev, plan = st.upload(imgs)
for _ in range(5):
    out = st.run(dev, plan)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(args.iters):
    out = st.run(dev, plan)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / args.iters
wr = out.numel() * 4
rd = sum(i.raw.size for i in imgs)
print(f'batch {args.batch} -> {tuple(out.shape)}: {ms * 1e3:.1f} us/launch, write {wr / 1e6:.1f} MB + read {rd / 1e6:.1f} MB '
      f'= {(wr + rd) / ms / 1e6:.0f} GB/s; PCIe bytes {rd / 1e6:.1f} MB vs {wr / 1e6:.1f} MB for host-side fp32 batches')
