"""world_size-2 gloo rehearsal (CPU) of the data-parallel gradient exchange: bucketed all-reduce of the flat
gradient buffer from autograd hooks, parameters without gradient contribute zeros, update divides by world."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from htd_amd.runner import FlatParams, GradientExchange
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    unused = torch.nn.Linear(3, 3)                     # never receives a gradient (graph_lvl{i}_cls of an empty level)
    holder = torch.nn.ModuleList([model, unused])
    flat = FlatParams(holder, bucket_mb=0)             # tiny buckets: several collectives per step
    ex = GradientExchange(flat)
    assert ex.enabled and len(flat.buckets) > 1
    x = torch.full((5, 8), float(rank + 1))
    flat.zero_grad()
    ex.begin_step()
    model(x).sum().backward()
    ex.finish_step()
    # reference: sum over ranks of the local gradients
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    tot = [torch.zeros_like(p) for p in ref.parameters()]
    for r in range(world):
        ref.zero_grad()
        ref(torch.full((5, 8), float(r + 1))).sum().backward()
        tot = [t + p.grad for t, p in zip(tot, ref.parameters())]
    ok = all(torch.allclose(p.grad, t, atol=1e-5) for p, t in zip(model.parameters(), tot))
    ok = ok and all(float(p.grad.abs().sum()) == 0.0 for p in unused.parameters())
    ok = ok and all(p.grad.data_ptr() >= flat.grad.data_ptr() for p in holder.parameters())   # still views of the flat buffer
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gradient_exchange_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_parse_losses_packs_one_allreduce():
    from htd_amd.detector.two_stage import BaseDetector
    det = BaseDetector()
    losses = {'loss_a': torch.tensor(1.5), 'loss_b': [torch.tensor(0.5), torch.tensor(1.0)], 'acc': torch.tensor(90.)}
    loss, log_vars = det._parse_losses(losses)
    assert abs(float(loss) - 3.0) < 1e-6                       # 'acc' is logged, not summed (base.py:212-213)
    assert list(log_vars.keys()) == ['loss_a', 'loss_b', 'acc', 'loss']
    assert abs(log_vars['loss_b'] - 1.5) < 1e-6 and abs(log_vars['loss'] - 3.0) < 1e-6
