#!/usr/bin/env python3
"""List the source lines of the train step that synchronise host and device (torch sync debug mode)."""
import os, sys, warnings, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
for _ in range(2):
    tr.train_step(data)
torch.cuda.synchronize()
hits = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if '/htd_amd/' in f.filename or f.filename.endswith('bench.py')]
    key = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st[-3:]))
    if not key:
        key = 'outside htd_amd: ' + ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in reversed(traceback.extract_stack()[-8:-1]))
    hits[key] += 1
warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
tr.train_step(data)
torch.cuda.set_sync_debug_mode('default')
for k, v in hits.most_common():
    print(v, k)
