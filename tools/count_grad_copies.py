import sys, os
sys.path.insert(0, '/root/repo')
import torch
from htd_amd.configs import build_htd_detector
from htd_amd.runner import Trainer, synthetic_batch, FlatParams
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_htd_detector(50).to(dev).train()
tr = Trainer(model)
data = synthetic_batch(4, device=dev)
for _ in range(2):
    tr.train_step(data)
names = {id(p): n for n, p in model.named_parameters()}
copied = []
orig = FlatParams.collect_one
def patched(self, i):
    p, v = self.params[i], self.grad_views[i]
    g = p.grad
    if g is not None and g.data_ptr() != v.data_ptr():
        copied.append((names.get(id(p), '?'), tuple(p.shape)))
    return orig(self, i)
FlatParams.collect_one = patched
tr.train_step(data)
torch.cuda.synchronize()
print('copies', len(copied), 'of', len(tr.flat.params))
for n, s in copied:
    print(n, s)
