"""Minimal training runner for the HTD path: what mmcv's EpochBasedRunner + OptimizerHook +
MMDistributedDataParallel do around `model.train_step` (mmdet/apis/train.py:72-150,
configs/_base_/schedules/schedule_1x.py:2-11), re-designed for one process per MI355X:

  * all trainable parameters live in ONE flat fp32 buffer (params are views into it), gradients in a
    second flat buffer, SGD momentum in a third: the optimizer is one fused HIP kernel over the flat
    buffers (htd_sgd_momentum_step), LR comes from a device scalar so warm-up needs no re-launch setup;
  * gradient exchange = bucketed RCCL all-reduce over xGMI of slices of the flat gradient buffer,
    launched from autograd post-accumulate hooks on a side stream while backward is still running
    (semantics of DDP as configured at apis/train.py:76-80 and of core/utils/dist_utils.py:10-51:
    sum then divide by world size; the division is folded into the optimizer kernel);
  * parameters that receive no gradient in a step (a graph_lvl{i}_cls whose level had no RoI,
    htd_bbox_head.py:219) contribute zeros: the flat gradient buffer is zero-filled each step.
"""
import math
import os

import torch
import torch.distributed as dist

from . import dense
from . import mmcv_ops as M

CL = torch.channels_last


class FlatParams:
    """Flattens the trainable parameters of `model` into contiguous fp32 buffers (16-byte aligned slices)."""

    def __init__(self, model, bucket_mb=64):
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.offsets, self.total = offs, total
        self.flat = torch.zeros(total, device=dev)
        self.grad = torch.zeros(total, device=dev)
        self.momentum = torch.zeros(total, device=dev)
        for p, o in zip(self.params, offs):
            self._view(self.flat, p, o).copy_(p.data)
            p.data = self._view(self.flat, p, o)
            p.grad = self._view(self.grad, p, o)
        # buckets in reverse parameter order (roughly the order backward produces gradients)
        cap = bucket_mb * (1 << 20) // 4
        self.buckets = []            # [lo, hi) slices of the flat buffers
        self.bucket_of = {}
        hi = total
        lo = total
        members = []
        for i in range(len(self.params) - 1, -1, -1):
            lo = offs[i]
            members.append(i)
            if hi - lo >= cap or i == 0:
                b = len(self.buckets)
                self.buckets.append((lo, hi))
                for m in members:
                    self.bucket_of[m] = b
                members, hi = [], lo
        self.grad_views = [self._view(self.grad, p, o) for p, o in zip(self.params, offs)]
        self._sink_used = set()
        self._sink_keys = dense.register_grad_sinks(self.params, self.grad_views, self._sink_used)

    def close(self):
        """Withdraw this buffer's gradient sinks (also runs when the object is collected)."""
        dense.unregister_grad_sinks(getattr(self, '_sink_keys', ()), getattr(self, '_sink_used', None))
        self._sink_keys = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _view(flat, p, off):
        n = p.numel()
        if p.dim() == 4 and p.is_contiguous(memory_format=CL) and not p.is_contiguous():
            co, ci, kh, kw = p.shape
            return flat[off:off + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat[off:off + n].view(p.shape)

    def zero_grad(self):
        """Zero the flat gradient buffer and detach .grad: the kernels that produce parameter gradients write them
        straight into their slice (dense.grad_out) and autograd adopts that tensor; collect() repairs the rest."""
        self.grad.zero_()
        dense.reset_grad_sinks(self._sink_used)
        for p in self.params:
            p.grad = None

    def collect_one(self, i):
        p, v = self.params[i], self.grad_views[i]
        g = p.grad
        if g is not None and g.data_ptr() != v.data_ptr():
            v.copy_(g)
        p.grad = v

    def collect(self):
        """After backward: every .grad is its slice of the flat buffer again (missing gradients = zeros)."""
        for i in range(len(self.params)):
            self.collect_one(i)


class GradientExchange:
    """Bucketed all-reduce of FlatParams.grad overlapped with backward (RCCL when the process group is nccl)."""

    def __init__(self, flat, process_group=None, comm_dtype=None):
        """comm_dtype=torch.bfloat16: buckets cross the links as bf16 (BASELINE configs[2]/[3]: 186.8 MB instead of
        373.6 MB for R101; what mmcv's Fp16OptimizerHook does with fp16 gradients) -- cast, all-reduce, cast back into
        the fp32 flat buffer; the sum over ranks is then accurate to bf16 (8 bits), the master gradients stay fp32."""
        self.flat = flat
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # a single-rank group still runs the exchange when HTD_REHEARSE_RCCL=1 (bench.py: one-GPU rehearsal of the
        # nccl code path)
        self.enabled = self.world > 1 or (dist.is_initialized() and os.environ.get('HTD_REHEARSE_RCCL') == '1')
        self.on_gpu = flat.grad.is_cuda
        self.stream = torch.cuda.Stream() if (self.enabled and self.on_gpu) else None
        self._pending = None
        self._works = []
        # ONE persistent payload buffer for the reduced-precision exchange (a bucket is cast into its slice, reduced there and
        # cast back): no allocation per bucket, and the allocator never sees the comm stream
        self._low = torch.empty_like(flat.grad, dtype=self.comm_dtype) if (self.enabled and self.comm_dtype is not None) else None
        # communication account for bench.py's N > 1 line: payload bytes per step and the EXPOSED part of the exchange -- from
        # the end of backward on the compute stream to the end of the last bucket on the comm stream (device events; wall clock
        # for a host-side backend).  Sampled on the steps between stats_begin() and stats().
        self._stat = None
        if self.enabled:
            for i, p in enumerate(flat.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(param):
            if self._pending is None:
                return
            self.flat.collect_one(i)
            b = self.flat.bucket_of[i]
            self._pending[b].discard(i)
            self._advance()
        return hook

    def _advance(self):
        """Launch ready buckets strictly in bucket order.  Every rank therefore issues the same sequence of
        collectives even when a data-dependent branch leaves some parameters without gradient on one rank only
        (their bucket then waits for finish_step on every rank that does have it ready -- order is what matters)."""
        while self._next < len(self.flat.buckets) and not self._pending[self._next]:
            self._launch(self._next)
            self._next += 1

    def begin_step(self):
        if not self.enabled:
            return
        self._pending = [set() for _ in self.flat.buckets]
        for i, b in self.flat.bucket_of.items():
            self._pending[b].add(i)
        self._launched = set()
        self._works = []
        self._next = 0

    def _launch(self, b):
        self._launched.add(b)
        lo, hi = self.flat.buckets[b]
        chunk = self.flat.grad[lo:hi]
        low = self._low[lo:hi] if self._low is not None else None
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            if dense._SIDE:
                self.stream.wait_stream(dense.side_stream(chunk.device))
            with torch.cuda.stream(self.stream):
                if low is None:
                    dist.all_reduce(chunk, group=self.group)
                else:
                    low.copy_(chunk)
                    dist.all_reduce(low, group=self.group)
                    chunk.copy_(low)
        elif low is None:
            self._works.append(dist.all_reduce(chunk, group=self.group, async_op=True))
        else:
            low.copy_(chunk)
            dist.all_reduce(low, group=self.group)
            chunk.copy_(low)

    # ---- communication account -------------------------------------------------------------------------------------
    def stats_begin(self):
        self._stat = dict(steps=0, events=[], host_ms=0.0)

    def stats(self):
        """-> dict(bytes_per_step, buckets, exposed_ms, ...) over the steps since stats_begin() (None when nothing is exchanged)."""
        st, self._stat = self._stat, None
        if not self.enabled or st is None or st['steps'] == 0:
            return None
        if st['events']:
            torch.cuda.synchronize()
            exposed = sum(max(0.0, a.elapsed_time(b)) for a, b in st['events']) / len(st['events'])
            how = 'device events: end of backward (compute stream) -> end of the last bucket (comm stream)'
        else:
            exposed = st['host_ms'] / st['steps']
            how = 'host clock around finish_step (host-side backend)'
        item = 2 if self.comm_dtype is not None else 4
        return dict(bytes_per_step=int(self.flat.grad.numel()) * item, buckets=len(self.flat.buckets),
                    payload_dtype=str(self.comm_dtype or torch.float32).replace('torch.', ''),
                    exposed_ms=round(exposed, 4), exposed_how=how, steps_sampled=st['steps'],
                    backend=dist.get_backend(self.group) if dist.is_initialized() else None, n_ranks=self.world)

    def finish_step(self):
        """Reduce the remaining buckets (parameters that got no gradient this step contribute zeros), still in
        bucket order, then join the side stream."""
        dense.join_side_stream()         # weight gradients are produced on dense's second stream
        self.flat.collect()              # every .grad is its flat slice again (unused parameters: zeros)
        if not self.enabled:
            return
        st = self._stat
        t0 = ev0 = None
        if st is not None:
            if self.stream is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()             # backward (and its weight-gradient stream) ends here on the compute stream
            else:
                import time
                t0 = time.perf_counter()
        while self._next < len(self.flat.buckets):
            self._launch(self._next)
            self._next += 1
        for w in self._works:
            w.wait()
        if self.stream is not None:
            if ev0 is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record(self.stream)  # the last bucket is done here on the comm stream
                st['events'].append((ev0, ev1))
            torch.cuda.current_stream().wait_stream(self.stream)
        if st is not None:
            st['steps'] += 1
            if t0 is not None:
                import time
                st['host_ms'] += (time.perf_counter() - t0) * 1e3
        self._pending = None


class WarmupStepLR:
    """lr policy 'step' with linear warm-up (schedule_1x.py:5-10): warmup_iters=500, warmup_ratio=0.001.
    `iters_per_epoch` = len(dataset) / (world * samples_per_gpu) places the decay points; 7330 is COCO train2017
    (117 266 images after filtering) at the reference's 8 x 2 images."""

    def __init__(self, base_lr, steps=(8, 11), gamma=0.1, warmup_iters=500, warmup_ratio=0.001, iters_per_epoch=7330):
        self.base_lr, self.steps, self.gamma = base_lr, tuple(steps), gamma
        self.warmup_iters, self.warmup_ratio, self.iters_per_epoch = warmup_iters, warmup_ratio, iters_per_epoch

    @classmethod
    def from_cfg(cls, cfg, iters_per_epoch):
        """From the reference's config keys: cfg.optimizer.lr, cfg.lr_config{policy, step, gamma, warmup, warmup_iters,
        warmup_ratio} (configs/_base_/schedules/schedule_1x.py:2-10, configs/htd/htd_resnet101_2x.py:119-127: step
        [16, 22] for the 2x configs).  iters_per_epoch must be given: it depends on the data set and the global batch."""
        lc = cfg['lr_config']
        if lc.get('policy', 'step') != 'step':
            raise NotImplementedError(f"lr policy {lc.get('policy')!r}: the HTD configs use 'step'")
        if lc.get('warmup', 'linear') not in ('linear', None):
            raise NotImplementedError(f"warmup {lc.get('warmup')!r}: the HTD configs use 'linear'")
        step = lc['step']
        return cls(cfg['optimizer']['lr'], steps=[step] if isinstance(step, int) else list(step), gamma=lc.get('gamma', 0.1),
                   warmup_iters=lc.get('warmup_iters', 0) if lc.get('warmup') else 0,
                   warmup_ratio=lc.get('warmup_ratio', 0.1), iters_per_epoch=int(iters_per_epoch))

    def lr(self, it):
        epoch = it // self.iters_per_epoch
        lr = self.base_lr * self.gamma ** sum(epoch >= s for s in self.steps)
        if it < self.warmup_iters:
            k = (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
            lr = lr * (1 - k)
        return lr


class Trainer:
    """train_step loop: zero_grad -> losses = model(**data) -> _parse_losses -> backward (+ overlapped gradient
    all-reduce) -> fused SGD update.  `data` = dict(img, img_metas, gt_bboxes, gt_labels)."""

    def __init__(self, model, lr=0.02, momentum=0.9, weight_decay=1e-4, schedule=None, bucket_mb=64, cfg=None,
                 iters_per_epoch=None, comm_dtype=None):
        """cfg (+ iters_per_epoch): take lr / momentum / weight_decay / schedule from the reference's config keys
        (cfg.optimizer, cfg.lr_config) instead of the keyword defaults."""
        if cfg is not None:
            opt = cfg['optimizer']
            if opt.get('type', 'SGD') != 'SGD':
                raise NotImplementedError(f"optimizer {opt.get('type')!r}: the HTD configs use SGD")
            lr, momentum, weight_decay = opt['lr'], opt.get('momentum', 0.0), opt.get('weight_decay', 0.0)
            if schedule is None:
                if iters_per_epoch is None:
                    raise ValueError('Trainer(cfg=...) needs iters_per_epoch = len(dataset) / (world * samples_per_gpu)')
                schedule = WarmupStepLR.from_cfg(cfg, iters_per_epoch)
        self.model = model
        self.flat = FlatParams(model, bucket_mb)
        self.exchange = GradientExchange(self.flat, comm_dtype=comm_dtype)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.schedule = schedule or WarmupStepLR(lr, iters_per_epoch=iters_per_epoch or 7330)
        self.lr_dev = torch.zeros(1, device=self.flat.flat.device)
        self.iter = 0
        self.sync_from_rank0()

    def sync_from_rank0(self):
        """What DistributedDataParallel does at construction (apis/train.py:72-80): every rank starts from rank 0's
        parameters and buffers.  Gradients are averaged, so ranks that start apart (different seed, a checkpoint
        loaded on rank 0 only) would otherwise stay apart silently."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        dist.broadcast(self.flat.flat, src=0)
        frozen = [p for p in self.model.parameters() if not p.requires_grad]
        for t in frozen + list(self.model.buffers()):
            if t.numel():
                if t.is_contiguous():
                    dist.broadcast(t.data, src=0)
                else:                                       # channels_last conv weights: dense in memory, not "contiguous"
                    buf = t.data.contiguous()
                    dist.broadcast(buf, src=0)
                    t.data.copy_(buf)
        M.PARAM_EPOCH += 1                                  # folded-weight caches see new parameters

    # ------------------------------------------------------------------ optimizer state / resume (apis/train.py:146-149)
    @property
    def epoch(self):
        return self.iter // self.schedule.iters_per_epoch

    def _numbering(self):
        """Index of every trainable parameter in the optimizer's numbering.  mmcv's DefaultOptimizerConstructor hands
        `model.parameters()` -- frozen ones included -- to torch.optim.SGD (apis/train.py:86), so the indices of a
        reference checkpoint count ALL parameters in module order; frozen ones simply never get a state entry."""
        pos = {id(p): i for i, p in enumerate(self.model.parameters())}
        return [pos[id(p)] for p in self.flat.params], len(pos)

    def state_dict(self):
        """torch.optim.SGD-shaped state (what mmcv's CheckpointHook stores under 'optimizer'): momentum buffers are
        per-parameter tensors in the parameter's logical shape, numbered like the reference's optimizer."""
        idx, n_all = self._numbering()
        # wire shape of a parameter = the shape of its state_dict entry (a TileLinear weight is (out, C*h*w) in files and
        # (out, C, h, w) channels_last in memory): momentum buffers are written in the shape torch.optim.SGD would hold
        name_of = {id(p): n for n, p in self.model.named_parameters()}
        wire = {n: tuple(v.shape) for n, v in self.model.state_dict().items()}
        state = {}
        for i, p, o in zip(idx, self.flat.params, self.flat.offsets):
            buf = FlatParams._view(self.flat.momentum, p, o).detach().clone().contiguous()
            state[i] = {'momentum_buffer': buf.reshape(wire.get(name_of.get(id(p)), tuple(p.shape)))}
        group = dict(lr=self.schedule.lr(self.iter), momentum=self.momentum, dampening=0, weight_decay=self.weight_decay,
                     nesterov=False, initial_lr=self.schedule.base_lr, params=list(range(n_all)))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        idx, n_all = self._numbering()
        groups = sd.get('param_groups', [])
        n = sum(len(g['params']) for g in groups)
        if n == len(idx) and n != n_all:
            idx = list(range(n))                            # a file that numbered the trainable parameters only
        elif n != n_all:
            raise ValueError(f'optimizer state numbers {n} parameters, the model has {n_all} ({len(idx)} trainable)')
        if groups:
            self.momentum = groups[0].get('momentum', self.momentum)
            self.weight_decay = groups[0].get('weight_decay', self.weight_decay)
        where = {i: k for k, i in enumerate(idx)}
        self.flat.momentum.zero_()
        for i, st in sd.get('state', {}).items():
            buf = st.get('momentum_buffer')
            if buf is None:
                continue
            if int(i) not in where:
                raise ValueError(f'optimizer state entry {int(i)} belongs to a parameter that is frozen here')
            k = where[int(i)]
            p, o = self.flat.params[k], self.flat.offsets[k]
            if tuple(buf.shape) != tuple(p.shape):
                # the reference's logical shape of a layer stored in another physical layout here (TileLinear: (out, C*h*w)
                # against (out, C, h, w) channels_last): same element order logically, so a view; anything else is an error
                if buf.dim() == 2 and p.dim() == 4 and buf.size(0) == p.size(0) and buf.numel() == p.numel():
                    buf = buf.reshape(p.shape)
                else:
                    raise ValueError(f'momentum buffer {int(i)}: shape {tuple(buf.shape)} vs parameter {tuple(p.shape)}')
            FlatParams._view(self.flat.momentum, p, o).copy_(buf)

    def save_checkpoint(self, filename, meta=None):
        """CheckpointHook's file: {'meta': {epoch, iter, ...}, 'state_dict', 'optimizer'}."""
        from .checkpoint import save_checkpoint
        meta = dict(meta or {})
        meta.update(epoch=self.epoch, iter=self.iter)
        return save_checkpoint(self.model, filename, optimizer=self.state_dict(), meta=meta)

    def resume(self, filename, map_location='cpu'):
        """runner.resume (apis/train.py:146-147): weights, optimizer state, epoch and iteration -- the LR schedule
        continues where it stopped (no second warm-up)."""
        from .checkpoint import load_checkpoint
        ckpt = load_checkpoint(self.model, filename, map_location=map_location, strict=True)
        if 'optimizer' in ckpt:
            self.load_state_dict(ckpt['optimizer'])
        meta = ckpt.get('meta', {})
        self.iter = int(meta.get('iter', meta.get('epoch', 0) * self.schedule.iters_per_epoch))
        M.PARAM_EPOCH += 1
        return ckpt

    def load_checkpoint(self, filename, map_location='cpu', strict=False):
        """runner.load_checkpoint (cfg.load_from, apis/train.py:148-149): weights only, training starts at iteration 0."""
        from .checkpoint import load_checkpoint
        ckpt = load_checkpoint(self.model, filename, map_location=map_location, strict=strict)
        M.PARAM_EPOCH += 1
        return ckpt

    def train_step(self, data):
        self.flat.zero_grad()
        self.exchange.begin_step()
        with dense.overlap_wgrad(getattr(self.model, 'overlap_wgrad', None)):
            out = self.model.train_step(data, None)
            out['loss'].backward()
        self.exchange.finish_step()
        self.lr_dev.fill_(self.schedule.lr(self.iter))
        if self.flat.flat.is_cuda:
            M.sgd_momentum_step_(self.flat.flat, self.flat.grad, self.flat.momentum, self.lr_dev, self.momentum,
                                 self.weight_decay, grad_scale=1.0 / self.exchange.world)
            dense.new_step()        # the kernel rewrote the weights in place: this step's flipped images are stale
        else:   # gloo / CPU rehearsal of the distributed logic only (tests): same arithmetic in torch
            g = self.flat.grad / self.exchange.world + self.weight_decay * self.flat.flat
            self.flat.momentum.mul_(self.momentum).add_(g)
            self.flat.flat.add_(self.flat.momentum, alpha=-float(self.lr_dev))
        self.iter += 1
        return out


def synthetic_batch(B, H=800, W=1344, img_w=1333, device='cuda', seed=0, num_classes=80):
    """COCO-shaped synthetic batch: randn images, 1-9 random gt boxes per image (recipe of
    the reference fixture tests/test_models/test_forward.py:311-328), labels in [0, 80)."""
    import numpy as np
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(1000 + seed)
    img = torch.randn(B, 3, H, W, generator=g)
    gts, labels = [], []
    for _ in range(B):
        k = rng.randint(1, 10)
        cx, cy, bw, bh = rng.rand(k, 4).T
        x1 = ((cx * img_w) - (img_w * bw / 2)).clip(0, img_w)
        y1 = ((cy * H) - (H * bh / 2)).clip(0, H)
        x2 = ((cx * img_w) + (img_w * bw / 2)).clip(0, img_w)
        y2 = ((cy * H) + (H * bh / 2)).clip(0, H)
        gts.append(torch.from_numpy(np.vstack([x1, y1, x2, y2]).T.astype(np.float32)))
        labels.append(torch.from_numpy(rng.randint(0, num_classes, size=k).astype(np.int64)))
    metas = [dict(img_shape=(H, img_w, 3), pad_shape=(H, W, 3), ori_shape=(H, img_w, 3),
                  scale_factor=np.array([1, 1, 1, 1], dtype=np.float32), flip=False) for _ in range(B)]
    dev = torch.device(device)
    return dict(img=img.to(dev).contiguous(memory_format=CL), img_metas=metas,
                gt_bboxes=[t.to(dev) for t in gts], gt_labels=[t.to(dev) for t in labels])
