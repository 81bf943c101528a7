#!/usr/bin/env python3
"""Device time of the ATen glue operators of one train step, grouped by operator and input shapes (torch.profiler)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

dev = torch.device('cuda:0')
torch.manual_seed(0)
if '--infer' in sys.argv:              # BASELINE configs[4]: R101, hard NMS, 512 proposals per image, batch INFER_B (64)
    from htd_amd.configs import htd_config
    cfg = htd_config(101, soft_nms=False)
    cfg.test_cfg.rpn.update(nms_post=512, max_num=512)
    model = build_htd_detector(cfg=cfg).to(dev).eval()
    data = synthetic_batch(int(os.environ.get('INFER_B', '64')), 800, 1344, 1333, device=dev)

    class tr:
        @staticmethod
        def train_step(d):
            with torch.no_grad():
                return model.simple_test(d['img'], d['img_metas'])
elif '--bf16' in sys.argv:           # BASELINE configs[2] per-GPU shape: HTD-R101, bf16 trunk / FC stacks
    model = build_htd_detector(101, bf16=True).to(dev).train()
    tr = Trainer(model, lr=0.015, comm_dtype=torch.bfloat16)
    data = synthetic_batch(4, device=dev)
else:
    model = build_htd_detector(50).to(dev).train()
    tr = Trainer(model)
    data = synthetic_batch(4, device=dev)
for _ in range(3):
    tr.train_step(data)
torch.cuda.synchronize()
STACK = '--stack' in sys.argv          # also print the Python frames of the top groups (which module launches them)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=STACK) as prof:
    tr.train_step(data)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6 if STACK else 0):
    if e.key.startswith('aten::') and e.self_device_time_total > 0:
        frames = [f for f in (getattr(e, 'stack', None) or []) if ('htd_amd' in f or 'bench' in f or 'tools/' in f) and 'aten_hot' not in f][:4]
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90], frames, e))
rows.sort(reverse=True, key=lambda r: r[0])
tot = sum(r[0] for r in rows)
print('ATen self device time: %.3f ms in %d op groups, %d launches' % (tot / 1e3, len(rows), sum(r[1] for r in rows)))
for t, n, k, sh, frames, e in rows[:int(os.environ.get('ATEN_TOP', '70'))]:
    print('%8.1f us %4d  %-28s %s' % (t, n, k, sh))
    for f in frames:
        print('              ' + f.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/', '')[:150])
    if STACK and not frames and t > 40:
        for f in (getattr(e, 'stack', None) or [])[:4]:
            print('              ? ' + f[:150])
