#!/usr/bin/env python3
"""tests/test_gpu_distributed.py::test_detector_world2_equals_averaged_gradients with names: the parameters whose update
differs between the two-rank run (gloo, one GPU) and the single-process sum of the ranks' gradients."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import copy
    from htd_amd import dense
    from htd_amd import mmcv_ops as M
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.core.bbox import set_sample_keys
    from htd_amd.runner import FlatParams, Trainer, synthetic_batch
    dev = torch.device('cuda:0')
    dense.OVERLAP_WGRAD = os.environ.get('DIAG_OVERLAP', '1') == '1'
    cfg = htd_config(50)
    cfg.train_cfg.rpn_proposal.update(nms_pre=300, nms_post=200, max_num=200)
    for r in cfg.train_cfg.rcnn:
        r.sampler.num = 64
    coef = torch.tensor([12.9898, 78.233, 37.719, 93.989], device=dev)
    set_sample_keys(lambda cand: torch.frac(torch.sin((cand * coef).sum(-1)) * 43758.5453).abs())
    torch.manual_seed(1)
    model = build_htd_detector(cfg=cfg).to(dev).train()
    tr = Trainer(model, lr=0.01, bucket_mb=8)
    start = tr.flat.flat.clone()
    ref_model = copy.deepcopy(model)
    datas = [synthetic_batch(2, 192, 256, 250, device=dev, seed=40 + r) for r in range(world)]
    steps = int(os.environ.get('DIAG_STEPS', '1'))
    for _ in range(steps):
        tr.train_step(datas[rank])
    torch.cuda.synchronize()
    got = tr.flat.flat.clone()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    offs, params = tr.flat.offsets, tr.flat.params
    tr.flat.close()
    flat = FlatParams(ref_model, bucket_mb=8)
    lr_dev = torch.zeros(1, device=dev)
    for it in range(steps):
        total = torch.zeros_like(flat.grad)
        for r in range(world):
            flat.zero_grad()
            ref_model.train_step(datas[r], None)['loss'].backward()
            dense.join_side_stream()
            flat.collect()
            total += flat.grad
        lr_dev.fill_(tr.schedule.lr(it))
        M.sgd_momentum_step_(flat.flat, total, flat.momentum, lr_dev, tr.momentum, tr.weight_decay, grad_scale=1.0 / world)
    torch.cuda.synchronize()
    if rank == 0:
        rows = []
        for n, o, p in zip(names, offs, params):
            a, b = (got - start)[o:o + p.numel()], (flat.flat - start)[o:o + p.numel()]
            rows.append((float((a - b).abs().max()), float(b.abs().max()), n, tr.flat.bucket_of[names.index(n)]))
        rows.sort(reverse=True)
        for e, s, n, b in rows[:14]:
            print(f'{n:58s} bucket {b:2d}  err {e:.3e}  step {s:.3e}')
        print('differing parameters:', sum(1 for e, *_ in rows if e > 0), 'of', len(rows), 'buckets', len(tr.flat.buckets))
    dist.destroy_process_group()


if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
