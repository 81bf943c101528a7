"""Host issue time of one train step against its device time.

Each step starts from an idle queue: `issue` is the host time until `train_step` returns (every launch queued, nothing
waited for unless the step itself waits), `done` the time until the device has drained.  issue ~ done means the host is
the bound (or the step blocks on a read-back); issue << done means the device is.

    python tools/host_issue.py [--depth 101] [--bf16] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--bf16', action='store_true')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--steps', type=int, default=10)
    args = ap.parse_args()
    from htd_amd import capi
    from htd_amd.configs import build_htd_detector
    from htd_amd.runner import Trainer, synthetic_batch
    capi.lib()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_htd_detector(args.depth, bf16=args.bf16).to(dev).train()
    trainer = Trainer(model, lr=0.02, comm_dtype=None)
    data = synthetic_batch(args.batch, 800, 1344, 1333, device=dev, seed=0)
    for _ in range(4):
        trainer.train_step(data)
    torch.cuda.synchronize()
    issue, done = [], []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        trainer.train_step(data)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3)
        done.append((t2 - t0) * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_step(data)
    torch.cuda.synchronize()
    back = (time.perf_counter() - t0) / args.steps * 1e3
    issue.sort()
    done.sort()
    print(f'depth {args.depth} bf16 {args.bf16}: issue median {issue[len(issue) // 2]:.2f} ms, done median '
          f'{done[len(done) // 2]:.2f} ms, back-to-back {back:.2f} ms per step')


if __name__ == '__main__':
    main()
