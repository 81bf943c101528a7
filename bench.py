#!/usr/bin/env python3
"""Headline benchmark: images/sec of the HTD-R50 train step (fwd + losses + bwd + gradient all-reduce +
SGD update) on synthetic 1333x800 COCO-shaped batches, fp32, B=4 per MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

N > 1 started plainly: this process touches no GPU, starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` as a CHILD (one rank per GPU over RCCL, the
launch of tools/dist_train.sh:7-9) and exits with its code.  Started under torch.distributed.run already (WORLD_SIZE
in the environment) it is a rank; WORLD_SIZE != --gpus is an error, never a silent one-GPU run.

Prints ONE JSON line on rank 0 (see the driver contract): whole-job images/sec, plus
  roofline     -- the dominant hand-written kernel of the step, timed live with device events on the stream
                  it is launched on, priced against its algorithmic FLOPs / bytes;
  cpu_baseline -- the CPU oracle (reference semantics) timed on this host on a bounded sample (rank 0, N=1).
Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0             # HBM3E peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X dense bf16 matrix peak (2:1 sparsity figures are not used)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4, help='images per GPU (BASELINE configs[1]: 4)')
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--bf16', action='store_true', help='bf16 backbone / FPN / RPN conv (BASELINE configs[2] precision)')
    ap.add_argument('--dcn', action='store_true', help='ResNet-DCN backbone (BASELINE configs[3] architecture, fp32 here)')
    ap.add_argument('--resnext', action='store_true', help='ResNeXt 64x4d backbone (htd_resnetx101_dcn_2x_mstrain.py with --depth 101 --dcn)')
    ap.add_argument('--pipeline', action='store_true', help='feed the step from uint8 images resident in HBM through the fused on-device '
                    'data pipeline (Resize 1333x800 keep-ratio / RandomFlip / Normalize / Pad 32) inside the timed region')
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1344)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--infer', action='store_true', help='inference-only throughput (BASELINE configs[4] shape): '
                    'simple_test with --proposals RoIs per image into the RoI head')
    ap.add_argument('--proposals', type=int, default=512)
    ap.add_argument('--profile-kernels', action='store_true', help='print the per-kernel-class time table')
    ap.add_argument('--profile-detail', action='store_true', help='per-layer-shape time table (implies the above)')
    ap.add_argument('--trained-like', action='store_true', help='force 128 stage-2 positives per image (what a trained '
                    'detector yields) so the HTD regression branch (1.24 GFLOP per positive RoI) is inside the timed region; '
                    'random-init weights give ~6 per image')
    ap.add_argument('--trained-like-steps', type=int, default=8, help='after the timed region: this many more timed steps with '
                    'trained-like proposals, reported as config.trained_like (0 = skip)')
    ap.add_argument('--cpu-baseline-full', action='store_true', help='ONLY the CPU baseline, by the BASELINE.md section 3 protocol '
                    '(3 warm-up + 10 timed steps, B=4 @ 800x1344); prints its JSON object; no GPU needed')
    ap.add_argument('--dry-launch', action='store_true', help='launcher rehearsal on CPU: ranks rendezvous over gloo, take the '
                    'barrier + MAX-over-ranks timing path and print the JSON line without touching a GPU (tests)')
    return ap.parse_args()


def trained_like_proposals(model, data, batch, per_gt_total=600):
    """--trained-like: a trained RPN puts many proposals on every object; a random-init one puts none, so stage 2 sees
    ~6 positives per image and the HTD regression branch (1.24 GFLOP per positive RoI) all but vanishes from the step.
    Here the LAST `per_gt_total` of the 2000 proposal slots of every image are overwritten, inside the step, by jittered
    copies of its gt boxes (IoU 0.65-0.95 with the gt): stage 1 then samples its full 128 positives (25 % of 512) and so
    does stage 2.  Everything else (NMS, sort, 2000 proposals per image) still runs."""
    rpn = model.rpn_head
    dev = data['img'].device
    g = torch.Generator().manual_seed(7)
    rows = []
    for b in range(batch):
        gt = data['gt_bboxes'][b].cpu()
        pick = gt[torch.randint(0, gt.size(0), (per_gt_total, ), generator=g)]
        wh = (pick[:, 2:] - pick[:, :2]).clamp(min=8.0)
        jit = (torch.rand(per_gt_total, 4, generator=g) - 0.5) * 0.12 * torch.cat([wh, wh], 1)
        rows.append(torch.cat([pick + jit, torch.full((per_gt_total, 1), 0.5)], 1))
    inject = torch.stack(rows).to(dev)                       # (B, J, 5)
    plain = rpn.get_bboxes

    def get_bboxes(*a, padded=False, **k):
        out = plain(*a, padded=padded, **k)
        if not padded:
            return out
        dets, n_keep = out
        dets = dets.clone()
        dets[:, -per_gt_total:] = inject
        return dets, torch.full_like(n_keep, dets.size(1))
    rpn.get_bboxes = get_bboxes


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child job and return its exit code.  Nothing in
    this process has initialised the GPU at this point (importing torch does not), and it never exec()s."""
    import socket
    import subprocess
    with socket.socket() as s:                  # a free rendezvous port on the loopback interface
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    print(f'# bench.py: starting {args.gpus} ranks: {" ".join(cmd)}', file=sys.stderr)
    return subprocess.call(cmd, env=env)


def cpu_baseline(depth, H, W, full=False, dcn=False, batch=4, infer=False, proposals=512):
    """The CPU oracle (oracle/detector.py, pinned against reference-generated fixtures): train step = forward + losses
    + backward, fp32, all host threads, the synthetic inputs of the GPU run.

    Default (every bench.py run): a BOUNDED sample -- one untimed warm-up step on a 256x320 image (thread pools, allocator,
    lazy kernels), then 2 timed steps on ONE 800x1344 image; value = images / mean step time, the two samples are
    reported.  full=True (`--cpu-baseline-full`, run once per round and kept under profiles/): the protocol of
    BASELINE.md section 3 / tools/benchmark.py:70-96 -- 3 warm-up + 10 timed steps at B = 4 @ 800x1344 (~10 minutes)."""
    from oracle import detector as D
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from golden_util import seeded_state_dict
    from htd_amd.runner import synthetic_batch
    threads = torch.get_num_threads()
    cfg = D.htd_config(depth, dcn)
    if infer:                                   # BASELINE configs[4] shape: hard NMS, `proposals` RoIs per image into the head
        cfg['test_cfg']['rpn'].update(nms_post=proposals, max_num=proposals)
        cfg['test_cfg']['rcnn']['nms'] = dict(type='nms', iou_threshold=0.5)
    sd = {k: v.requires_grad_(v.dtype.is_floating_point and 'running' not in k and not infer)
          for k, v in seeded_state_dict(D.state_shapes(depth, dcn), prefix='det.').items()}

    def step(data):
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        if infer:
            with torch.no_grad():
                D.simple_test(sd, data['img'].contiguous(), data['img_metas'], cfg)
            return time.perf_counter() - t0
        losses = D.forward_train(sd, data['img'].contiguous(), data['img_metas'], data['gt_bboxes'], data['gt_labels'], cfg)
        loss, _ = D.parse_losses(losses)
        loss.backward()
        return time.perf_counter() - t0

    torch.manual_seed(0)
    B, warm, timed = (batch, 3, 10) if full else (1, 1, 2)
    data = synthetic_batch(B, H, W, W - 11, device='cpu', seed=0)
    small = synthetic_batch(1, 256, 320, 309, device='cpu', seed=0)
    for i in range(warm):
        dt = step(data if full else small)
        print(f'# cpu_baseline warm-up {i + 1}/{warm}: {dt:.1f} s', file=sys.stderr, flush=True)
    ts = []
    for i in range(timed):
        ts.append(step(data))
        print(f'# cpu_baseline step {i + 1}/{timed}: {ts[-1]:.1f} s', file=sys.stderr, flush=True)
    mean = sum(ts) / len(ts)
    return dict(value=round(B / mean, 5), unit='images/sec', cores=threads, kind='port',
                sample=(f'oracle (CPU restatement of the reference path) ' + (f'simple_test ({proposals} proposals/img, hard NMS)' if infer else 'train step fwd+loss+bwd') + f', R{depth}{"-DCN" if dcn else ""} fp32, '
                        f'B={B} @{H}x{W}, {warm} warm-up + {timed} timed steps, mean {mean:.1f} s/step '
                        f'(min {min(ts):.1f}, max {max(ts):.1f})' +
                        ('' if full else '; bounded sample -- the BASELINE.md section 3 protocol (3 + 10 steps at B=4) is '
                                         'run by --cpu-baseline-full and kept in profiles/')))


def pipeline_batch(batch, dev, rank):
    """600x1000 uint8 images (-> 800x1333, padded 800x1344: the shape of the default workload) planned by the reference's
    train pipeline, uploaded once; feed() runs the fused kernel on the resident bytes."""
    import numpy as np
    from htd_amd.pipelines import DeviceBatchStager, build_pipeline, collate
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    pipe = build_pipeline([dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
                           dict(type='Resize', img_scale=(1333, 800), keep_ratio=True),
                           dict(type='RandomFlip', flip_ratio=0.5), dict(type='Normalize', **norm),
                           dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'),
                           dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])])
    rng = np.random.RandomState(rank)
    np.random.seed(rank)
    samples = []
    for i in range(batch):
        k = rng.randint(1, 10)
        cx, cy, bw, bh = rng.rand(k, 4).T
        boxes = np.stack([(cx - bw / 2).clip(0, 1) * 1000, (cy - bh / 2).clip(0, 1) * 600, (cx + bw / 2).clip(0, 1) * 1000,
                          (cy + bh / 2).clip(0, 1) * 600], 1).astype(np.float32)
        samples.append(pipe(dict(img=rng.randint(0, 256, (600, 1000, 3)).astype(np.uint8), img_prefix=None, bbox_fields=[],
                                 img_info=dict(filename=f'{i}.jpg'),
                                 ann_info=dict(bboxes=boxes, labels=rng.randint(0, 80, k).astype(np.int64)))))
    data = collate(samples, dev)
    stager = DeviceBatchStager(dev)
    resident, plan = stager.upload([s['img'] for s in samples])
    return data, lambda: stager.run(resident, plan)


def dry_launch(args, world, rank):
    """The rank protocol of the real run without a GPU: gloo rendezvous, barrier, timed region, MAX over ranks, rank 0
    prints the line.  n_gpus is the world size the process group reports, not the flag."""
    if world > 1:
        dist.init_process_group('gloo')
    seen = dist.get_world_size() if dist.is_initialized() else 1
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank == 0:
        print(json.dumps({'metric': 'launcher rehearsal (no GPU work)', 'value': 0.0, 'unit': 'images/sec', 'n_gpus': seen,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed * 1e3, 3),
                          'config': {'workload': 'dry launch', 'global_batch': args.batch * seen, 'parallelism': f'dp{seen}'}}))


def main():
    args = parse()
    launched = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a '
                         f'{args.gpus}-GPU number from {world} rank(s)')
    if args.dry_launch:
        return dry_launch(args, world, rank)
    if args.cpu_baseline_full:
        print(json.dumps(cpu_baseline(args.depth, args.height, args.width, full=True, dcn=args.dcn, batch=args.batch,
                                      infer=args.infer, proposals=args.proposals)))
        return
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (the HIP ops have no CPU path)'
    # HTD_BENCH_BACKEND=gloo HTD_BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a ONE-GPU box (tests/test_gpu_distributed.py):
    # every rank runs the real detector on cuda:0 and the exchange goes over gloo instead of RCCL -- rank-0 broadcast, autograd
    # hooks + side stream, packed log all-reduce, barrier, MAX-over-ranks timing and the JSON line are the code of the real run.
    backend = os.environ.get('HTD_BENCH_BACKEND', 'nccl')
    share_gpu = backend == 'gloo' and bool(int(os.environ.get('HTD_BENCH_SHARE_GPU', '0')))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # HTD_REHEARSE_RCCL=1: initialise RCCL and run the bucketed gradient exchange even with ONE rank -- the only way
    # to exercise the nccl-backend code path (side stream, hooks, packed log all-reduce) on a one-GPU box
    rehearse = bool(int(os.environ.get('HTD_REHEARSE_RCCL', '0')))
    if world > 1 or rehearse:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)     # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    n_ranks = dist.get_world_size() if dist.is_initialized() else 1      # what RCCL actually sees
    if n_ranks != args.gpus and not rehearse:
        raise SystemExit(f'bench.py: process group has {n_ranks} ranks, --gpus says {args.gpus}')

    from htd_amd import capi
    from htd_amd.configs import build_htd_detector
    from htd_amd.runner import Trainer, synthetic_batch
    capi.lib()                                              # fail loudly if the HIP library is missing

    torch.manual_seed(0)
    if args.infer:
        from htd_amd.configs import htd_config
        cfg = htd_config(args.depth, dcn=args.dcn, soft_nms=False, resnext=args.resnext)   # configs[4]: hard NMS, 512 proposals
        cfg.test_cfg.rpn.update(nms_post=args.proposals, max_num=args.proposals)
        model = build_htd_detector(cfg=cfg, bf16=args.bf16).to(dev).eval()
        data = synthetic_batch(args.batch, args.height, args.width, args.width - 11, device=dev, seed=rank)

        class _Infer:                                       # same .train_step() shape as Trainer for the loop below
            def train_step(self, d):
                with torch.no_grad():
                    return model.simple_test(d['img'], d['img_metas'])
        trainer = _Infer()
    else:
        model = build_htd_detector(args.depth, dcn=args.dcn, bf16=args.bf16, resnext=args.resnext)  # init_weights(), seed 0
        model = model.to(dev).train()
        # bf16 configurations: gradients cross xGMI as bf16 (186.8 MB for R101 instead of 373.6 MB), fp32 master buffers
        trainer = Trainer(model, lr=0.02 if args.depth == 50 else 0.015, comm_dtype=torch.bfloat16 if args.bf16 else None)
        data = synthetic_batch(args.batch, args.height, args.width, args.width - 11, device=dev, seed=rank)
        if args.trained_like:
            trained_like_proposals(model, data, args.batch)
        if args.pipeline:
            data, feed = pipeline_batch(args.batch, dev, rank)
            step_plain = trainer.train_step

            def step_with_pipeline(d):                      # one launch of htd_image_batch_pipeline per step
                d['img'] = feed()
                return step_plain(d)
            trainer.train_step = step_with_pipeline

    def sync():
        if world > 1 or rehearse:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(data)
    sync()
    # live device-event timing of the matrix-core entry points (the roofline kernels); every entry point with
    # --profile-kernels (costs ~6 % of the step)
    everything = args.profile_kernels or args.profile_detail
    capi.profile_begin(detail=args.profile_detail, only=None if everything else (
        'htd_conv2d_fwd', 'htd_conv2d_bwd_data', 'htd_conv2d_fwd_x3p', 'htd_conv2d_bwd_data_x3p', 'htd_conv2d_bwd_weight',
        'htd_conv2d_fwd_x3h', 'htd_conv2d_bwd_data_x3h', 'htd_conv2d_bwd_weight_h2',
        'htd_bgemm_nt', 'htd_conv2d_fwd_bf16',
        'htd_conv2d_dgrad_bf16', 'htd_conv2d_bwd_weight_bf16'))
    exchange = getattr(trainer, 'exchange', None)
    if exchange is not None:
        exchange.stats_begin()
    npos_headline = None
    t0 = time.perf_counter()
    for i in range(args.steps):
        # kernel events on every 4th step of the timed region (all steps with --profile-kernels): their queue
        # barriers cost ~2 ms per timed step, which would otherwise be charged to the throughput being measured
        capi.profile_pause(not everything and i % 4 != 0)
        trainer.train_step(data)
    sync()
    elapsed = time.perf_counter() - t0
    print(f'# timed region: {elapsed:.3f} s for {args.steps} steps', file=sys.stderr)
    prof = capi.profile_end()
    comm = exchange.stats() if exchange is not None else None
    if not args.infer and hasattr(model.roi_head, '_last_static'):
        npos_headline = int(model.roi_head._last_static[1].npos.sum())
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        gathered = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([elapsed], device=dev, dtype=torch.float64))
        rank_ms = [round(float(g) / args.steps * 1e3, 3) for g in gathered]
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- the HTD workload a TRAINED detector sees, inside the same driver-timed process: with random-init
    # weights stage 2 samples ~28 positives per batch and the regression branch (htd_bbox_head.py:77-113, 1.24 GFLOP per
    # positive RoI) all but vanishes from the step above.  A few more steps with jittered gt boxes injected into the proposal
    # lists (both stages then sample their 128 positives per image); reported beside the headline, never as `value`.
    trained = None
    if not args.infer and not args.trained_like and args.trained_like_steps > 0:
        trained_like_proposals(model, data, args.batch)
        TL_WARMUP = 4          # (the regression branch's tensors are new sizes for the allocator: two steps did not always cover them)
        for _ in range(TL_WARMUP):
            trainer.train_step(data)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.trained_like_steps):
            trainer.train_step(data)
        sync()
        el = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        npos_t = int(model.roi_head._last_static[1].npos.sum()) if hasattr(model.roi_head, '_last_static') else None
        trained = dict(img_s=round(args.batch * n_ranks * args.trained_like_steps / el, 3),
                       ms_per_step=round(el / args.trained_like_steps * 1e3, 3), steps=args.trained_like_steps, warmup=TL_WARMUP,
                       stage2_positives=npos_t, reg_branch_gflop=round(npos_t * 1.2355, 1) if npos_t is not None else None,
                       proposals='jittered gt boxes injected into the last 600 proposal slots of every image: 128 positives / '
                                 'image / stage (bench.py trained_like_proposals)')
    if rank != 0:
        return
    ms = elapsed / args.steps * 1e3
    world = n_ranks
    value = args.batch * world * args.steps / elapsed

    from htd_amd import dense
    # fp32 convolutions run as six bf16 MFMA products per fp32 product (exact three-way bf16 splits of both operands,
    # fp32 accumulation: conv_fwd.hip "X3"): the roof of that algorithm is the bf16 matrix peak / 6 in algorithmic fp32
    # FLOP/s; with HTD_CONV_MATH=0 (fp32-input MFMA) it is the fp32 matrix peak
    x3 = capi.lib().htd_conv2d_set_math(-1) == 1
    roof = dense.roofline_report(prof, PEAK_BF16_MFMA_TFLOPS / 6.0 if x3 else PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS,
                                 PEAK_BF16_MFMA_TFLOPS)
    if roof and roof.get('bound') == 'mfma' and 'bf16' not in roof['kernel']:
        roof['peak'] = round(roof['peak'], 1)
        h2 = 'H2 form' in roof['kernel']
        if h2:
            roof['math'] = ('fp32 = 3 x v_mfma_f32_32x32x16_f16 on two block-scaled fp16 pieces per operand (a0 b0 + a0 b1 + a1 b0), fp32 '
                            'accumulate; peak = 2500 / 3 algorithmic TFLOP/s.  Under the socket power cap the matrix pipe alone sustains '
                            '539 algorithmic TFLOP/s in this form and 284 in the six-product bf16 form (tools/micro/mfma_split_products.hip, '
                            'profiles/r04_mfma_split_products.txt)')
            roof['power_capped_matrix_pipe_tflops'] = 539.2
            roof['frac_of_power_capped_matrix_pipe'] = round(roof['achieved'] / 539.2, 4)
            roof['frac_of_six_product_bf16_roof'] = round(roof['achieved'] / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4)    # rounds 2-3's yardstick
        else:
            roof['math'] = ('fp32 = 6 x v_mfma_f32_32x32x16_bf16 on exact 3-way bf16 splits, fp32 accumulate; peak = 2500 / 6 '
                            'algorithmic TFLOP/s (the fp32-input MFMA peak, 157.3, is not the bound of this kernel)') if x3 else \
                'v_mfma_f32_32x32x2_f32'
        roof['issued_bf16_tflops'] = round(roof['achieved'] * (3.0 if h2 else 6.0), 1) if x3 else None
        roof['frac_of_fp32_input_mfma_peak'] = round(roof['achieved'] / PEAK_F32_MFMA_TFLOPS, 4)      # 157.3: round 1's yardstick
    # HBM bytes per launch of the dominant kernel class come from separate rocprofv3 PMC passes (FETCH_SIZE,
    # WRITE_SIZE; gfx950 read correction applied) whose summary is committed under profiles/
    try:
        # passes taken on the kernels of that arithmetic: the newest round that has them
        names = ['r04_hbm_traffic.json', 'r03_hbm_traffic.json', 'r02_hbm_traffic.json'] if x3 else ['r01_hbm_traffic.json']
        cls = 'conv_wgrad' if 'wgrad' in roof['kernel'] else ('conv_x3p' if 'x3p' in roof['kernel'] else 'conv_igemm')
        for name in names:
            path = os.path.join(ROOT, 'profiles', name)
            tr = json.load(open(path))['kernels'] if os.path.exists(path) else {}
            if cls in tr and 'conv' in roof['kernel'] and 'bf16' not in roof['kernel'] and args.depth == 50 and not args.infer \
                    and args.batch == 4 and not args.trained_like:
                roof['traffic'] = tr[cls]['hbm_bytes_per_launch_corrected']
                roof['traffic_unit'] = f'bytes/launch (rocprofv3 PMC passes of this command, profiles/{name})'
                break
    except Exception:
        pass
    if args.profile_kernels or args.profile_detail:
        for name, (n, tot_ms, kind, work, _) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:60]:
            rate = (work / (tot_ms * 1e-3) / 1e12) if (kind and tot_ms > 0) else 0.0
            print(f'# {name:64s} calls={n:5d} total={tot_ms:9.3f} ms  {kind or ""} {rate:8.2f} T/s', file=sys.stderr)
    out = {
        'metric': ('images/sec (1333x800) HTD-%s%d inference' if args.infer else
                   'images/sec (1333x800) HTD-%s%d train step') % ('X' if args.resnext else 'R', args.depth),
        'value': round(value, 3),
        'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16' if args.bf16 else 'f32',        # tensors, accumulation and results fp32; see roofline.math for the products
        'data': 'synthetic uint8 images through the on-device data pipeline' if args.pipeline else 'synthetic',
        'config': {'workload': (f'HTD {"ResNeXt-64x4d" if args.resnext else "ResNet"}-{args.depth}{"-DCN" if args.dcn else ""} FPN {"bf16 (backbone / FPN / RPN conv / RoI FC stacks)" if args.bf16 else "fp32"} inference (simple_test, hard NMS), batch {args.batch}/GPU '
                                f'@ {args.width - 11}x{args.height}, {args.proposals} proposals/img into the RoI head'
                                if args.infer else
                                f'HTD {"ResNeXt-64x4d" if args.resnext else "ResNet"}-{args.depth}{"-DCN" if args.dcn else ""} FPN {"bf16 (backbone / FPN / RPN conv / RoI FC stacks; fp32 master weights, RoI ops, losses)" if args.bf16 else "fp32"} train step fwd+bwd+SGD, batch {args.batch}/GPU @ '
                                f'{args.width - 11}x{args.height} (padded {args.width}x{args.height}), '
                                'random-init weights, 2000 RPN proposals/img, 512 RoIs/img/stage'),
                   'global_batch': args.batch * world, 'parallelism': f'dp{world}'},
        'roofline': roof,
    }
    if backend != 'nccl' and world > 1:
        out['rehearsal'] = f'{backend} backend' + (f', {world} ranks sharing cuda:0' if share_gpu else '') + \
            ': exercises the rank protocol, NOT a multi-GPU measurement'

    if not args.infer and hasattr(model.roi_head, '_last_static'):
        # how much of the HTD regression branch (3x3 256->576->576->576->1024 on 7x7, htd_bbox_head.py:77-113) the timed
        # step contained: it runs on stage-2 positives only (1.2355 GFLOP forward per positive RoI)
        npos = npos_headline
        out['config']['stage2_positives'] = npos
        out['config']['reg_branch_gflop'] = round(npos * 1.2355, 1)
        out['config']['proposals'] = 'trained-like (jittered gt boxes injected: 128 positives/img/stage)' if args.trained_like \
            else 'random-init RPN (few positives: the regression branch is nearly idle)'
    if trained is not None:
        out['config']['trained_like'] = trained
    if world > 1 or comm is not None:
        # what RCCL saw and what the exchange cost, so that a sub-linear point of the scaling curve can be attributed
        try:
            ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
        out['comm'] = dict(comm or {}, rccl_version=ver, process_group_ranks=n_ranks)
        out['ms_per_step_per_rank'] = rank_ms
    if world == 1 and not args.no_cpu_baseline and not args.infer:
        out['cpu_baseline'] = cpu_baseline(args.depth, args.height, args.width, dcn=args.dcn)
    print(json.dumps(out))


if __name__ == '__main__':
    try:
        main()
    finally:
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
