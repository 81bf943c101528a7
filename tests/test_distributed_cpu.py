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


def _worker(rank, world, port, q, comm_dtype=None):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from htd_amd.runner import FlatParams, GradientExchange
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    unused = torch.nn.Linear(3, 3)                     # never receives a gradient (graph_lvl{i}_cls of an empty level)
    holder = torch.nn.ModuleList([model, unused])
    flat = FlatParams(holder, bucket_mb=0)             # tiny buckets: several collectives per step
    ex = GradientExchange(flat, comm_dtype=comm_dtype)
    assert ex.enabled and len(flat.buckets) > 1
    x = torch.full((5, 8), float(rank + 1))
    flat.zero_grad()
    ex.stats_begin()
    ex.begin_step()
    model(x).sum().backward()
    ex.finish_step()
    st = ex.stats()          # bench.py's `comm` object: payload bytes, buckets, exposed time, ranks seen
    item = 4 if comm_dtype is None else 2
    assert st['bytes_per_step'] == flat.grad.numel() * item and st['buckets'] == len(flat.buckets) and st['n_ranks'] == world
    assert st['steps_sampled'] == 1 and st['exposed_ms'] >= 0.0 and st['backend'] == 'gloo'
    assert ex.stats() is None                          # sampling ended with the read
    # reference: sum over ranks of the local gradients
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    tot = [torch.zeros_like(p) for p in ref.parameters()]
    for r in range(world):
        ref.zero_grad()
        ref(torch.full((5, 8), float(r + 1))).sum().backward()
        tot = [t + p.grad for t, p in zip(tot, ref.parameters())]
    # bf16 payload: every rank's bucket is rounded to bf16 (8 bits) before the sum
    tol = dict(atol=1e-5) if comm_dtype is None else dict(rtol=2 ** -7, atol=1e-3)
    ok = all(torch.allclose(p.grad, t, **tol) for p, t in zip(model.parameters(), tot))
    ok = ok and all(float(p.grad.abs().sum()) == 0.0 for p in unused.parameters())
    ok = ok and all(p.grad.data_ptr() >= flat.grad.data_ptr() for p in holder.parameters())   # still views of the flat buffer
    q.put((rank, ok))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('comm_dtype', [None, torch.bfloat16], ids=['fp32', 'bf16_payload'])
def test_gradient_exchange_world2(comm_dtype):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, comm_dtype)) for r in range(world)]
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


def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from htd_amd.runner import Trainer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(6, 5)
            self.bn = torch.nn.BatchNorm1d(5)
            self.frozen = torch.nn.Linear(5, 2)
            for p in self.frozen.parameters():
                p.requires_grad_(False)

        def train_step(self, data, optimizer):
            return dict(loss=self.frozen(self.bn(self.a(data))).pow(2).mean())

    torch.manual_seed(100 + rank)                          # ranks start APART
    net = Net()
    net.bn.running_mean.fill_(float(rank + 1))
    tr = Trainer(net, lr=0.1)                              # broadcasts rank 0's parameters and buffers (DDP construction)
    gathered = [torch.zeros_like(tr.flat.flat) for _ in range(world)]
    dist.all_gather(gathered, tr.flat.flat)
    ok = all(torch.equal(g, gathered[0]) for g in gathered)
    for t in (net.frozen.weight.data, net.bn.running_mean):
        g = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(g, t.contiguous())
        ok = ok and all(torch.equal(x, g[0]) for x in g)
    # different data per rank, two steps: parameters stay identical across ranks (averaged gradients)
    torch.manual_seed(rank)
    for _ in range(2):
        tr.train_step(torch.randn(7, 6))
    dist.all_gather(gathered, tr.flat.flat)
    ok = ok and all(torch.allclose(g, gathered[0], atol=0, rtol=0) for g in gathered)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_trainer_broadcasts_rank0_state_at_construction():
    """ranks that start from different weights must not stay different (DDP broadcasts at wrap time,
    apis/train.py:72-80)."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` started plainly launches 2 ranks (tools/dist_train.sh:7-9), reports the world size the
    process group saw, and refuses a rank count that differs from --gpus.  --dry-launch: gloo, no GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-launch'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                        # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['config']['global_batch'] == 8
    # one rank but --gpus 2: an error, not a silent one-GPU run
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry-launch'],
                       env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'refusing' in r.stderr
    # N = 1: same line shape, no launcher
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--dry-launch'], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])['n_gpus'] == 1
