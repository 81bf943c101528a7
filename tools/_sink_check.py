import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
tr.train_step(data)
names = {id(p): n for n, p in model.named_parameters()}
stat = {'in_place': 0, 'copied': [], 'none': []}
orig = tr.flat.collect_one
def co(i):
    p, v = tr.flat.params[i], tr.flat.grad_views[i]
    g = p.grad
    if g is None: stat['none'].append(names[id(p)])
    elif g.data_ptr() == v.data_ptr(): stat['in_place'] += 1
    else: stat['copied'].append(names[id(p)])
    orig(i)
tr.flat.collect_one = co
tr.train_step(data)
print('in place', stat['in_place'], 'copied', len(stat['copied']), 'none', len(stat['none']))
print('copied:', stat['copied'][:60])
print('none:', stat['none'][:20])
