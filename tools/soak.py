#!/usr/bin/env python3
"""Soak run of the headline train step: time per step and allocator state every CHUNK steps (a leak or a slow drift shows up as
growing `reserved`), with the shader clock / power / temperature `rocm-smi` reports at that moment; after the run a REST
seconds pause and one more chunk tell a thermal drift (recovers) from a software one (does not).
The workload itself drifts when the detector learns the four synthetic batches: more proposals become stage-2 positives and
the regression branch (1.24 GFLOP per positive RoI and pass) grows; SOAK_LR=0 keeps the weights fixed.
usage: [SOAK_LR=0] python tools/soak.py [steps] [chunk] [rest_seconds]"""
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 250
rest = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lr = float(os.environ.get('SOAK_LR', '0.02'))      # 0: the weights stay, so does the workload (number of positives)


def smi():
    try:
        r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--csv'], capture_output=True, text=True, timeout=20)
        rows = [l for l in r.stdout.splitlines() if l.strip()]
        head, vals = rows[0].split(','), rows[1].split(',')
        keep = [(h, v) for h, v in zip(head, vals) if re.search(r'sclk|Power|junction|edge', h)]
        return '  '.join(f'{h.strip()}={v.strip()}' for h, v in keep)[:260]
    except Exception as e:          # the tool is informational
        return f'(rocm-smi: {e})'
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
tr = Trainer(model, lr=lr)
datas = [synthetic_batch(4, device=dev, seed=s) for s in range(4)]
for i in range(5):
    tr.train_step(datas[i % 4])
torch.cuda.synchronize()
print('after warm-up: allocated %.1f MB, reserved %.1f MB' % (torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6), flush=True)
t0 = time.perf_counter()
for i in range(steps):
    out = tr.train_step(datas[i % 4])
    last_loss = out['loss'].detach()
    del out
    if (i + 1) % chunk == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        from htd_amd import dense
        if dense.H2_GUARD:              # HTD_H2_GUARD=1: no H2 launch of the chunk's last step overflowed its split (dense.h2_check)
            dense.h2_check()
        print('             loss %.4f%s' % (float(last_loss), '  (H2 guard clean)' if dense.H2_GUARD else ''), flush=True)
        print('steps %5d: %.2f ms/step in this chunk, allocated %.1f MB, peak %.1f MB, reserved %.1f MB' %
              (i + 1, (t1 - t0) / chunk * 1e3, torch.cuda.memory_allocated() / 1e6, torch.cuda.max_memory_allocated() / 1e6,
               torch.cuda.memory_reserved() / 1e6), flush=True)
        print('             stage-1 / stage-2 positives of the last step: %d / %d   ' % tuple(int(S.npos.sum()) for S in model.roi_head._last_static) + smi(), flush=True)
        t0 = time.perf_counter()
if rest:
    time.sleep(rest)
    print(f'rested {rest} s:  ' + smi(), flush=True)
    t0 = time.perf_counter()
    for i in range(chunk):
        tr.train_step(datas[i % 4])
    torch.cuda.synchronize()
    print('after the rest: %.2f ms/step over %d steps' % ((time.perf_counter() - t0) / chunk * 1e3, chunk), flush=True)
