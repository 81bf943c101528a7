#!/usr/bin/env python3
"""Per-step wall time of the headline train step (host clock around train_step + synchronize) with the garbage collector's
pauses beside it: which steps are slow, and is it the collector?     usage: python tools/step_jitter.py [steps]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
pauses, t_gc = [], [0.0]


def cb(phase, info):
    if phase == 'start':
        t_gc[0] = time.perf_counter()
    else:
        pauses.append((info['generation'], (time.perf_counter() - t_gc[0]) * 1e3))


gc.callbacks.append(cb)
for _ in range(5):
    tr.train_step(data)
torch.cuda.synchronize()
rows = []
for i in range(steps):
    n0 = len(pauses)
    t0 = time.perf_counter()
    tr.train_step(data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, pauses[n0:]))
for i, (issue, total, ps) in enumerate(rows):
    print('step %3d  issue %6.2f ms  total %6.2f ms  %s' % (i, issue, total, ' '.join('gc%d:%.1fms' % p for p in ps)))
tot = sorted(r[1] for r in rows)
print('median %.2f ms, min %.2f, max %.2f; gc pauses: %d, %.1f ms in all' % (tot[len(tot) // 2], tot[0], tot[-1], len(pauses),
                                                                          sum(p[1] for p in pauses)))
