"""Two ranks sharing the one GPU of the test box (gloo backend, CUDA tensors): the GPU code path of the gradient
exchange -- autograd hooks, side stream, in-order buckets, fused SGD on the flat buffers -- with world_size 2.
(RCCL itself needs one GPU per rank; the driver exercises it in the multi-GPU bench.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.b = torch.nn.Linear(32, 8)
        self.unused = torch.nn.Linear(4, 4)

    def train_step(self, data, optimizer):
        return dict(loss=self.b(torch.relu(self.a(data['x']))).pow(2).mean())


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from htd_amd.runner import Trainer
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = Tiny().to(dev)
    tr = Trainer(model, lr=0.1, momentum=0.9, weight_decay=1e-4, bucket_mb=0)
    tr.schedule.warmup_iters = 0
    assert tr.exchange.enabled and tr.exchange.stream is not None and len(tr.flat.buckets) > 1
    x = torch.full((6, 16), 0.1 * (rank + 1), device=dev)
    for _ in range(3):
        tr.train_step(dict(x=x))
    torch.cuda.synchronize()
    # reference: plain SGD on the rank-averaged gradient
    torch.manual_seed(0)
    ref = Tiny().to(dev)
    opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    for _ in range(3):
        opt.zero_grad()
        for r in range(world):
            (ref.train_step(dict(x=torch.full((6, 16), 0.1 * (r + 1), device=dev)), None)['loss'] / world).backward()
        for p in ref.unused.parameters():
            p.grad = torch.zeros_like(p)
        opt.step()
    ok = all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(model.parameters(), ref.parameters()))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_trainer_world2_on_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


@pytest.mark.gpu
def test_weight_gradient_stream_gives_identical_step():
    """dense.OVERLAP_WGRAD queues weight gradients on a second HIP stream; one optimizer step must leave the parameters of
    the single-stream run (same kernels, same order per tensor)."""
    import copy
    from htd_amd import dense
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.runner import Trainer, synthetic_batch
    dev = torch.device('cuda:0')
    cfg = htd_config(50)
    cfg.train_cfg.rpn_proposal.update(nms_pre=300, nms_post=200, max_num=200)
    for r in cfg.train_cfg.rcnn:
        r.sampler.num = 64
    torch.manual_seed(0)
    base = build_htd_detector(cfg=cfg).to(dev).train()
    data = synthetic_batch(2, 256, 320, 311, device=dev, seed=3)
    flats = []
    saved = dense.OVERLAP_WGRAD
    try:
        for overlap in (False, True):
            dense.OVERLAP_WGRAD = overlap
            model = copy.deepcopy(base)
            tr = Trainer(model, lr=0.01)
            torch.manual_seed(11)
            tr.train_step(data)
            torch.cuda.synchronize()
            flats.append(tr.flat.flat.clone())
    finally:
        dense.OVERLAP_WGRAD = saved
    assert torch.isfinite(flats[0]).all()
    # float atomics of the RoIAlign backward make the gradients reproducible to rounding only
    torch.testing.assert_close(flats[0], flats[1], rtol=0, atol=1e-6)
