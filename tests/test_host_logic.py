"""CPU tests (no GPU): the C-ABI library loads and exports every symbol of include/htd_amd.h, the registry /
config surface builds the HTD detector with the reference's state_dict keys, and the product's device-agnostic
host logic (anchors, IoU, assigner, sampler, coder, losses, batched PGraph) reproduces the reference-generated
fixtures and the CPU oracle."""
import ctypes
import os

import numpy as np
import pytest
import torch

from golden_util import seeded_tensor, seeded_state_value

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_capi_exports_every_declared_symbol():
    from htd_amd import capi
    from htd_amd.csrc import build as hip_build
    hip_build.build()
    decl = capi.declared_functions()
    assert len(decl) >= 20
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name, _, _ in decl:
        assert hasattr(lib, name), f'{name} declared in include/htd_amd.h but not exported'
    capi.lib()
    assert capi.lib().htd_abi_version() >= 1
    # argument errors come back as status + message, never abort (no GPU needed: rejected before any launch)
    with pytest.raises(ValueError):
        capi.call('htd_roi_align_fwd', None, None, None, 0, None, 5, 1, 3, 4, 4, 7, 7, 1.0, 0, 1, None)


def test_ops_refuse_cpu_tensors():
    from htd_amd import dense, mmcv_ops as M
    with pytest.raises(NotImplementedError):
        M.roi_align(torch.zeros(1, 4, 4, 4), torch.zeros(1, 5), 7)
    with pytest.raises(NotImplementedError):
        M.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)
    with pytest.raises(NotImplementedError):
        dense.linear(torch.zeros(2, 8), torch.zeros(4, 8))


def test_registry_and_config_build_detector():
    from htd_amd import Config, ConfigDict
    from htd_amd.configs import build_htd_detector, htd_config
    from oracle import detector as D
    for depth in (50, 101):
        det = build_htd_detector(depth)
        mine = {k: tuple(v.shape) for k, v in det.state_dict().items() if not k.endswith('num_batches_tracked')}
        ref = {k: tuple(v) for k, v in D.state_shapes(depth).items()}
        aliases = {k for k in mine if '.att.' in k}         # AdptRoIExtractor.att.{1,3} alias conv1/conv2 (as in the reference)
        assert set(mine) - aliases == set(ref)
        assert all(mine[k] == ref[k] for k in ref)
    n_all = sum(p.numel() for p in det.parameters())
    n_train = sum(p.numel() for p in det.parameters() if p.requires_grad)
    assert abs(n_all / 1e6 - 93.63) < 0.01 and abs(n_train / 1e6 - 93.40) < 0.01
    cfg = htd_config(50)
    assert isinstance(cfg.train_cfg.rcnn[1], ConfigDict) and cfg.train_cfg.rcnn[1].assigner.pos_iou_thr == 0.6
    assert cfg.train_cfg.get('rpn_proposal').nms_post == 2000


def test_config_fromfile_with_base(tmp_path):
    from htd_amd import Config
    (tmp_path / 'base.py').write_text("model = dict(type='FasterRCNN', backbone=dict(type='ResNet', depth=50))\nlr = 0.02\n")
    (tmp_path / 'child.py').write_text("_base_ = ['./base.py']\nmodel = dict(backbone=dict(depth=101))\n")
    cfg = Config.fromfile(str(tmp_path / 'child.py'))
    assert cfg.model.backbone.depth == 101 and cfg.model.backbone.type == 'ResNet' and cfg.lr == 0.02
    cfg.merge_from_dict({'model.backbone.depth': 50})
    assert cfg.model.backbone.depth == 50


def test_reference_config_file_loads_if_present():
    """The reference's own config file builds through this registry (authoring container only)."""
    path = '/root/reference/configs/htd/htd_resnet50_1x.py'
    if not os.path.exists(path):
        pytest.skip('reference tree not present on this machine')
    from htd_amd import Config, build_detector
    from htd_amd import detector  # noqa: F401
    cfg = Config.fromfile(path)
    model = cfg.model.to_dict()
    model['pretrained'] = None
    det = build_detector(model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(det).__name__ == 'FasterRCNN' and type(det.roi_head).__name__ == 'HTDRoIHead'


def test_anchors_exact(golden):
    from htd_amd.core import AnchorGenerator
    g = golden('anchors')
    ag = AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    sizes = [tuple(s) for s in g['sizes']]
    anchors = ag.grid_anchors(sizes, device='cpu')
    flags = ag.valid_flags(sizes, tuple(g['pad_shape']), device='cpu')
    for i in range(5):
        assert torch.equal(anchors[i], T(g[f'anchors{i}'])) and torch.equal(flags[i], T(g[f'flags{i}']))
    kat = AnchorGenerator([10], [1.], [1.], [10]).grid_anchors([(2, 2)], device='cpu')[0]    # tests/test_anchor.py:22-40
    assert torch.equal(kat, T(g['kat_anchor']))
    assert torch.equal(AnchorGenerator([(10, 20)], [1.], [1.], [10]).grid_anchors([(2, 2)], device='cpu')[0],
                       T(g['kat_anchor_xy']))


def test_box_math_and_assigner_and_sampler(golden):
    from htd_amd.core import (MaxIoUAssigner, RandomSampler, bbox2delta, bbox_overlaps, delta2bbox, set_randperm)
    g = golden('box_math')
    b1, b2 = T(g['b1']), T(g['b2'])
    assert torch.equal(bbox_overlaps(b1, b2), T(g['iou']))
    assert torch.equal(bbox_overlaps(b1, b2, mode='iof'), T(g['iof']))
    assert torch.equal(bbox_overlaps(b1[:9], b2, is_aligned=True), T(g['iou_aligned']))
    torch.testing.assert_close(bbox2delta(b1[:9], b2, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)), T(g['deltas']),
                               rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(delta2bbox(b1, T(g['rnd']), (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2), (80, 90, 3)),
                               T(g['dec']), rtol=1e-6, atol=1e-5)
    torch.testing.assert_close(delta2bbox(T(g['kat_rois']), T(g['kat_deltas']), max_shape=(32, 32)), T(g['kat_dec']),
                               rtol=1e-6, atol=1e-6)
    a = golden('assigner')
    r = MaxIoUAssigner(0.5, 0.5).assign(T(a['kat_bboxes']), T(a['kat_gts']), gt_labels=torch.LongTensor([2, 3]))
    assert r.gt_inds.tolist() == [1, 0, 2, 0]
    for tag, kw in [('rpn', dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)),
                    ('rcnn', dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False))]:
        r = MaxIoUAssigner(**kw).assign(T(a[f'{tag}_bboxes']), T(a[f'{tag}_gts']), gt_labels=T(a[f'{tag}_gt_labels']))
        assert torch.equal(r.gt_inds, T(a[f'{tag}_gt_inds'])) and torch.equal(r.labels, T(a[f'{tag}_labels']))
        assert torch.equal(r.max_overlaps, T(a[f'{tag}_max_overlaps']))
    assert (MaxIoUAssigner(0.5, 0.5).assign(T(a['kat_bboxes']), torch.empty(0, 4)).gt_inds == 0).all()
    assert MaxIoUAssigner(0.5, 0.5).assign(torch.empty(0, 4), T(a['kat_gts'])).gt_inds.numel() == 0
    # low-quality matching, gt_max_assign_all=False: last gt wins on a shared arg-max box (sequential overwrite)
    ov = torch.tensor([[0.4, 0.2, 0.0], [0.4, 0.1, 0.35]])
    r = MaxIoUAssigner(0.9, 0.3, min_pos_iou=0.3, gt_max_assign_all=False).assign_wrt_overlaps(ov)
    assert r.gt_inds.tolist() == [2, 0, -1]
    s = golden('sampler')
    pb, gb, gl = T(s['bboxes']), T(s['gts']), T(s['gt_labels'])
    ar = MaxIoUAssigner(0.3, 0.3, 0.3, match_low_quality=False).assign(pb, gb, gt_labels=gl)
    set_randperm(lambda n, dev: torch.randperm(n).to(dev))
    try:
        torch.manual_seed(int(s['seed']))
        res = RandomSampler(64, 0.25, -1, True).sample(ar, pb, gb, gl)
    finally:
        set_randperm(None)
    for k in ('pos_inds', 'neg_inds', 'pos_is_gt', 'pos_assigned_gt_inds', 'pos_gt_labels', 'pos_bboxes', 'neg_bboxes'):
        assert torch.equal(getattr(res, k), T(s[k])), k


def test_losses_match_oracle():
    from htd_amd.detector.losses import CrossEntropyLoss, SmoothL1Loss, accuracy
    from oracle import boxes as B
    g = torch.Generator().manual_seed(0)
    pred, lab, w = torch.randn(40, 81, generator=g), torch.randint(0, 81, (40, ), generator=g), torch.rand(40, generator=g)
    torch.testing.assert_close(CrossEntropyLoss()(pred, lab, w, avg_factor=17.), B.cross_entropy(pred, lab, w, 17.))
    p1, l1 = torch.randn(50, 1, generator=g), torch.randint(0, 2, (50, ), generator=g)
    torch.testing.assert_close(CrossEntropyLoss(use_sigmoid=True)(p1, l1, w.new_ones(50), avg_factor=9.),
                               B.binary_cross_entropy(p1, l1, w.new_ones(50), 9.))
    a, b, ww = torch.randn(30, 4, generator=g), torch.randn(30, 4, generator=g), torch.rand(30, 4, generator=g)
    torch.testing.assert_close(SmoothL1Loss(beta=1 / 9.)(a, b, ww, avg_factor=5.), B.smooth_l1_loss(a, b, ww, 1 / 9., 5.))
    torch.testing.assert_close(accuracy(pred, lab), B.accuracy(pred, lab))
    assert accuracy(torch.zeros(0, 81), torch.zeros(0, dtype=torch.long)).item() == 0.          # empty input


def test_batched_assign_and_sample_equals_per_image_path():
    """core.bbox.batched_assign_and_sample (one host read per stage) against MaxIoUAssigner + SamplingResult per
    image on the same picks: identical assignment / labels / boxes, counts follow the sampler's rules.  Includes an
    image without ground truth and ragged proposal / gt counts."""
    from htd_amd.core.bbox import (MaxIoUAssigner, RandomSampler, SamplingResult, batched_assign_and_sample)
    g = torch.Generator().manual_seed(0)

    def boxes(n, s=200.):
        xy = torch.rand(n, 2, generator=g) * s
        wh = torch.rand(n, 2, generator=g) * s * 0.4 + 2
        return torch.cat([xy, xy + wh], 1)
    gts = [boxes(5), boxes(0), boxes(2)]
    labels = [torch.randint(0, 80, (len(b), ), generator=g) for b in gts]
    props = [torch.cat([boxes(300), gts[0] + 1.0]), boxes(150), torch.cat([boxes(40), gts[2] + 0.5, gts[2] - 0.5])]
    assigner = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)
    sampler = RandomSampler(num=64, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    res, counts = batched_assign_and_sample(assigner, sampler, props, gts, labels)
    for b in range(3):
        r = res[b]
        ar = assigner.assign(props[b], gts[b], None, labels[b])
        cand = props[b]
        flags = cand.new_zeros((cand.shape[0], ), dtype=torch.uint8)
        if len(gts[b]) > 0:
            cand = torch.cat([gts[b], cand], 0)
            ar.add_gt_(labels[b])
            flags = torch.cat([cand.new_ones(len(gts[b]), dtype=torch.uint8), flags])
        n_cand_pos, n_cand_neg = int((ar.gt_inds > 0).sum()), int((ar.gt_inds == 0).sum())
        assert counts[b][0] == min(n_cand_pos, 16) and counts[b][1] == min(n_cand_neg, 64 - counts[b][0])
        assert (ar.gt_inds[r.pos_inds] > 0).all() and (ar.gt_inds[r.neg_inds] == 0).all()
        assert torch.equal(torch.sort(r.pos_inds)[0], r.pos_inds) and torch.equal(torch.sort(r.neg_inds)[0], r.neg_inds)
        ref = SamplingResult(r.pos_inds, r.neg_inds, cand, gts[b], ar, flags)
        for k in ('pos_bboxes', 'neg_bboxes', 'pos_is_gt', 'pos_assigned_gt_inds', 'pos_gt_bboxes', 'pos_gt_labels'):
            assert torch.equal(getattr(r, k), getattr(ref, k)), (b, k)
        assert int(r.pos_is_gt.sum()) == counts[b][2] and r.pos_is_gt[:counts[b][2]].all()     # gt rows lead


def test_checkpoint_wire_format_round_trip(tmp_path):
    """A checkpoint in the reference's format (state_dict with the reference's keys and logical shapes, `module.`
    prefix, meta, foreign keys) loads into the re-laid-out model (KRSC conv weights, (out,h,w,C) TileLinear) and a
    saved checkpoint has the reference's keys / shapes again."""
    from htd_amd.checkpoint import load_checkpoint, save_checkpoint
    from htd_amd.configs import build_htd_detector, htd_config
    from oracle import detector as D
    from golden_util import seeded_state_dict
    model = build_htd_detector(cfg=htd_config(50))
    ref = seeded_state_dict(D.state_shapes(50), prefix='ckpt.')
    ref = {k: torch.as_tensor(v) for k, v in ref.items()}
    ex = 'roi_head.bbox_roi_extractor.1.'           # a real reference file also holds the aliases att.1 / att.3
    for a, b in (('att.1', 'conv1'), ('att.3', 'conv2')):
        for t in ('weight', 'bias'):
            ref[f'{ex}{a}.{t}'] = ref[f'{ex}{b}.{t}']
    src = {'module.' + k: v for k, v in ref.items()}
    torch.save(dict(meta=dict(epoch=12), state_dict=src), tmp_path / 'ref.pth')
    ckpt = load_checkpoint(model, str(tmp_path / 'ref.pth'), strict=True)
    assert ckpt['meta']['epoch'] == 12
    w = model.roi_head.bbox_head[0].shared_fcs[0].weight                      # TileLinear: stored (out, C, h, w)
    assert w.shape == (1024, 256, 7, 7)
    assert torch.equal(w.reshape(1024, -1), ref['roi_head.bbox_head.0.shared_fcs.0.weight'])
    cw = model.backbone.layer2[0].conv2.weight
    assert cw.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(cw, ref['backbone.layer2.0.conv2.weight'])
    # foreign / missing keys: reported, not fatal unless strict
    partial = {k: v for k, v in ref.items() if k.startswith('backbone.')}
    bb = {k[len('backbone.'):]: v for k, v in partial.items() if k.startswith('backbone.')}
    bb['fc.weight'] = torch.zeros(3, 3)                                       # torchvision's classifier head
    torch.save(bb, tmp_path / 'resnet50.pth')
    model.backbone.init_weights(pretrained=str(tmp_path / 'resnet50.pth'))    # the reference's call site
    with pytest.raises(RuntimeError):
        load_checkpoint(model.backbone, str(tmp_path / 'resnet50.pth'), strict=True)
    with pytest.raises(IOError):
        load_checkpoint(model, 'torchvision://resnet50')
    save_checkpoint(model, str(tmp_path / 'out' / 'epoch_1.pth'), meta=dict(epoch=1))
    saved = torch.load(tmp_path / 'out' / 'epoch_1.pth', weights_only=False)
    got = {k: tuple(v.shape) for k, v in saved['state_dict'].items() if not k.endswith('num_batches_tracked')}
    assert got == {k: tuple(v.shape) for k, v in ref.items()}
    assert torch.equal(saved['state_dict']['roi_head.bbox_head.1.fcs.0.weight'], ref['roi_head.bbox_head.1.fcs.0.weight'])


def test_bbox_mapping_round_trip_and_aug_merge():
    """core/bbox/transforms.py:34-55 and merge_augs.py:54-81 (host tensors)."""
    import numpy as np
    import torch
    from htd_amd.core.bbox import bbox_flip, bbox_mapping, bbox_mapping_back
    from htd_amd.core.post_processing import merge_aug_bboxes, merge_aug_scores
    from oracle import detector as D
    boxes = torch.tensor([[10., 20., 50., 80.], [0., 0., 99., 59.]])
    meta = dict(img_shape=(75, 125, 3), scale_factor=np.array([1.25] * 4, dtype=np.float32), flip=True,
                flip_direction='horizontal')
    fwd = bbox_mapping(boxes, meta['img_shape'], meta['scale_factor'], True, 'horizontal')
    torch.testing.assert_close(fwd, torch.tensor([[62.5, 25., 112.5, 100.], [1.25, 0., 125., 73.75]]))
    torch.testing.assert_close(fwd, D.bbox_mapping(boxes, meta))
    torch.testing.assert_close(bbox_mapping_back(fwd, meta['img_shape'], meta['scale_factor'], True, 'horizontal'), boxes)
    torch.testing.assert_close(bbox_flip(bbox_flip(boxes, (60, 100), 'diagonal'), (60, 100), 'diagonal'), boxes)
    torch.testing.assert_close(bbox_flip(boxes, (60, 100), 'vertical')[0], torch.tensor([10., -20., 50., 40.]))
    # class-wise boxes (n, 4*classes) of two augmentations average after mapping back
    a = torch.tensor([[10., 10., 20., 20., 0., 0., 8., 8.]])
    b = bbox_mapping(a.view(-1, 4), meta['img_shape'], meta['scale_factor'], True, 'horizontal').view(1, 8)
    plain = dict(img_shape=(60, 100, 3), scale_factor=1.0, flip=False, flip_direction=None)
    merged, scores = merge_aug_bboxes([a, b], [torch.tensor([[.2, .8, 0.]]), torch.tensor([[.4, .4, .2]])],
                                      [[plain], [meta]], None)
    torch.testing.assert_close(merged, a)
    torch.testing.assert_close(scores, torch.tensor([[.3, .6, .1]]))
    torch.testing.assert_close(merge_aug_scores([torch.ones(2), torch.zeros(2)]), torch.full((2,), .5))


class _TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.frozen = torch.nn.Conv2d(3, 4, 3, padding=1)          # numbered by the reference's optimizer, never updated
        for p in self.frozen.parameters():
            p.requires_grad_(False)
        self.conv = torch.nn.Conv2d(4, 6, 3, padding=1).to(memory_format=torch.channels_last)
        self.fc = torch.nn.Linear(6, 2)

    def train_step(self, data, optimizer):
        y = self.conv(self.frozen(data)).mean((2, 3))
        return dict(loss=self.fc(y).pow(2).mean())


def test_trainer_state_dict_resume_continues_exactly(tmp_path):
    """ADVICE r1 / apis/train.py:146-149: a checkpoint carries momentum and iteration, resume continues the warm-up /
    step schedule where it stopped and reproduces the uninterrupted run bit for bit; the 'optimizer' entry has
    torch.optim.SGD's shape with the reference's numbering (frozen parameters counted, without state)."""
    from htd_amd.runner import Trainer, WarmupStepLR
    sched = lambda: WarmupStepLR(0.1, steps=(1, ), warmup_iters=4, warmup_ratio=0.1, iters_per_epoch=3)
    g = torch.Generator().manual_seed(0)
    batches = [torch.randn(2, 3, 5, 5, generator=g) for _ in range(6)]
    torch.manual_seed(1)
    a = Trainer(_TinyNet(), schedule=sched())
    for b in batches[:3]:
        a.train_step(b)
    a.save_checkpoint(str(tmp_path / 'epoch_1.pth'), meta=dict(CLASSES=('x', )))
    for b in batches[3:]:
        a.train_step(b)
    ck = torch.load(tmp_path / 'epoch_1.pth', weights_only=True)
    assert ck['meta']['iter'] == 3 and ck['meta']['epoch'] == 1
    opt = ck['optimizer']
    assert opt['param_groups'][0]['params'] == list(range(6)) and sorted(opt['state']) == [2, 3, 4, 5]
    assert opt['state'][2]['momentum_buffer'].shape == (6, 4, 3, 3)
    ref = torch.optim.SGD(_TinyNet().parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    ref.load_state_dict(opt)                                        # torch's own optimizer accepts the entry
    torch.manual_seed(2)                                            # different init: everything must come from the file
    b2 = Trainer(_TinyNet(), schedule=sched())
    b2.resume(str(tmp_path / 'epoch_1.pth'))
    assert b2.iter == 3 and b2.schedule.lr(b2.iter) == a.schedule.lr(3)
    for b in batches[3:]:
        b2.train_step(b)
    assert torch.equal(a.flat.flat, b2.flat.flat) and torch.equal(a.flat.momentum, b2.flat.momentum)
    # load_from semantics: weights only, iteration 0
    c = Trainer(_TinyNet(), schedule=sched())
    c.load_checkpoint(str(tmp_path / 'epoch_1.pth'))
    assert c.iter == 0 and float(c.flat.momentum.abs().sum()) == 0.0
    assert torch.equal(c.model.conv.weight, ck['state_dict']['conv.weight'])


def test_lr_schedule_comes_from_the_config():
    """R101 configs decay at epochs 16 / 22 of 24 (configs/htd/htd_resnet101_2x.py:119-127), R50 at 8 / 11."""
    from htd_amd.configs import htd_config
    from htd_amd.runner import Trainer, WarmupStepLR
    s50 = WarmupStepLR.from_cfg(htd_config(50), iters_per_epoch=100)
    s101 = WarmupStepLR.from_cfg(htd_config(101), iters_per_epoch=100)
    assert s50.base_lr == 0.02 and s101.base_lr == 0.015
    assert abs(s50.lr(0) - 0.02 * 0.001) < 1e-12 and abs(s50.lr(250) - 0.02 * (1 - 0.5 * 0.999)) < 1e-12
    assert s50.lr(799) == 0.02 and abs(s50.lr(800) - 0.002) < 1e-12 and abs(s50.lr(1100) - 0.0002) < 1e-12
    assert s101.lr(1599) == 0.015 and abs(s101.lr(1600) - 0.0015) < 1e-12 and abs(s101.lr(2200) - 0.00015) < 1e-12
    with pytest.raises(ValueError):
        Trainer(_TinyNet(), cfg=htd_config(101))                    # iters_per_epoch is not guessable
    t = Trainer(_TinyNet(), cfg=htd_config(101), iters_per_epoch=50)
    assert t.schedule.steps == (16, 22) and t.schedule.iters_per_epoch == 50 and t.weight_decay == 0.0001


def test_checkpoint_loader_rules(tmp_path, monkeypatch):
    """mmcv 1.2.1's prefix rule (first key decides), safe unpickling by default, pretrained:// names from a local dir."""
    from htd_amd.checkpoint import load_checkpoint
    net = torch.nn.Linear(3, 2)
    sd = {'module.weight': torch.ones(2, 3), 'module.bias': torch.zeros(2)}
    torch.save(dict(state_dict=sd), tmp_path / 'a.pth')
    load_checkpoint(net, str(tmp_path / 'a.pth'), strict=True)
    assert float(net.weight.detach().sum()) == 6.0

    import pickle
    with open(tmp_path / 'b.pth', 'wb') as f:                       # a pickled foreign object: needs explicit trust
        pickle.dump(dict(state_dict={k: v.detach().numpy() for k, v in net.state_dict().items()},
                         meta=dict(obj=pytest.approx(1.0))), f)
    with pytest.raises(RuntimeError, match='trust'):
        load_checkpoint(net, str(tmp_path / 'b.pth'))
    # a merely BROKEN file keeps its own error -- it must not advertise the unsafe path, trusted or not
    whole = (tmp_path / 'a.pth').read_bytes()
    (tmp_path / 'c.pth').write_bytes(whole[:len(whole) // 2])
    for trusted in (False, True):
        with pytest.raises(Exception) as info:
            load_checkpoint(net, str(tmp_path / 'c.pth'), trusted=trusted)
        assert 'trust' not in str(info.value) and not isinstance(info.value, pickle.UnpicklingError)
    (tmp_path / 'zoo').mkdir()
    torch.save(net.state_dict(), tmp_path / 'zoo' / 'resnet50-19c8e357.pth')
    monkeypatch.setenv('HTD_PRETRAINED_DIR', str(tmp_path / 'zoo'))
    load_checkpoint(net, 'torchvision://resnet50', strict=True)
    with pytest.raises(IOError):
        load_checkpoint(net, 'torchvision://resnet101')


def test_gradient_sinks_are_scoped_to_their_trainer():
    """a second FlatParams must not disable the first one's sinks, and a dropped one must not leave entries
    that a later tensor at the same address could pick up."""
    import gc
    from htd_amd import dense
    from htd_amd.runner import FlatParams
    n0 = len(dense._GRAD_SINK)
    a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    fa, fb = FlatParams(a), FlatParams(b)
    assert len(dense._GRAD_SINK) == n0 + 4
    with torch.no_grad():
        ga, used_a = dense.grad_out2(a.weight)
        gb, used_b = dense.grad_out2(b.weight)
    assert used_a and used_b and ga.data_ptr() == fa.grad_views[0].data_ptr() and gb.data_ptr() == fb.grad_views[0].data_ptr()
    with torch.no_grad():
        assert not dense.grad_out2(a.weight)[1]                     # handed out once per step
    fa.zero_grad()
    with torch.no_grad():
        assert dense.grad_out2(a.weight)[1] and not dense.grad_out2(b.weight)[1]      # a's reset does not touch b's
    key = a.weight.data_ptr()
    del fa
    gc.collect()
    assert key not in dense._GRAD_SINK and len(dense._GRAD_SINK) == n0 + 2
    fb.close()
    assert len(dense._GRAD_SINK) == n0


def test_split_heads_backward_equals_slicing():
    """rpn_head._SplitHeads: channel slices of the merged 1x1 head; its one-concatenation backward equals what autograd
    computes for two plain slices (including a head that received no gradient)."""
    import torch
    from htd_amd.detector.rpn_head import _SplitHeads
    torch.manual_seed(0)
    y0 = torch.randn(2, 16, 5, 7).contiguous(memory_format=torch.channels_last)
    rc, rr = torch.randn(2, 3, 5, 7), torch.randn(2, 12, 5, 7)
    ya = y0.clone().requires_grad_()
    c, r = _SplitHeads.apply(ya, 3, 12)
    assert torch.equal(c, y0[:, :3]) and torch.equal(r, y0[:, 3:15])
    ((c * rc).sum() + (r * rr).sum()).backward()
    yb = y0.clone().requires_grad_()
    ((yb[:, :3] * rc).sum() + (yb[:, 3:15] * rr).sum()).backward()
    assert torch.equal(ya.grad, yb.grad) and float(ya.grad[:, 15:].abs().sum()) == 0.0
    ya = y0.clone().requires_grad_()
    c, r = _SplitHeads.apply(ya, 3, 12)
    (r * rr).sum().backward()                       # the classification slice unused
    assert float(ya.grad[:, :3].abs().sum()) == 0.0 and torch.equal(ya.grad[:, 3:15], rr)


def _x3p_plan(cfg, M, Co, Ci, k):
    from htd_amd import capi
    out = (ctypes.c_int64 * 8)()
    assert capi.lib().htd_conv2d_x3p_plan_query(cfg, M, Co, Ci, k, k, ctypes.cast(out, ctypes.c_void_p)) == 0
    return dict(zip(('tiles_a', 'splits_a', 'steps_a', 'splits_b', 'steps_b', 'm_rem0', 'grid', 'partial'), [int(v) for v in out]))


def test_x3p_work_plan_covers_every_tile_and_k_step_once():
    """conv_x3.hip plan_x3p (host code, no GPU): for the layer shapes of the headline step and a sweep of odd sizes, every
    tile is in exactly one region, the K ranges of a region tile cover all steps with no empty range, the grid is the sum of
    (tile, range) units, region B starts on a tile row, and the workspace the library asks for holds the plan's partial
    sums for whichever tile configuration the launch picks."""
    from htd_amd import capi
    tile = [(64, 64), (128, 128), (128, 64), (64, 128)]
    shapes = [(268800, 256, 256, 3), (67200, 256, 256, 3), (16800, 256, 256, 3), (4200, 512, 512, 3), (1092, 256, 256, 3),
              (16800, 1024, 256, 1), (16800, 256, 1024, 1), (4200, 2048, 512, 1), (4200, 512, 2048, 1), (67200, 512, 128, 1),
              (2048, 1024, 12544, 1), (2048, 12544, 1024, 1), (4096, 1024, 1024, 1), (1176, 576, 576, 3), (2048, 81, 1024, 1),
              (96, 256, 128, 1), (4, 81, 256, 1), (33333, 96, 48, 3), (257 * 64, 64, 64, 1), (255 * 128 + 1, 130, 4096, 1)]
    planned = 0
    for M, Co, Ci, k in shapes:
        need = capi.lib().htd_conv2d_x3p_workspace_bytes(M, Co, Ci, k, k)
        steps = (Ci // 16) * k
        for cfg, (bm, bn) in enumerate(tile):
            pl = _x3p_plan(cfg, M, Co, Ci, k)
            mt, nt = -(-M // bm), -(-Co // bn)
            assert 0 <= pl['tiles_a'] <= mt * nt and pl['tiles_a'] % nt == 0
            assert pl['m_rem0'] == min(pl['tiles_a'] // nt * bm, M)
            assert pl['grid'] == pl['tiles_a'] * pl['splits_a'] + (mt * nt - pl['tiles_a']) * pl['splits_b']
            for splits, sps in ((pl['splits_a'], pl['steps_a']), (pl['splits_b'], pl['steps_b'])):
                assert splits >= 1 and (splits - 1) * sps < steps <= splits * sps          # all steps, last range not empty
            partial = (pl['splits_a'] * pl['m_rem0'] * Co if pl['splits_a'] > 1 else 0) + \
                (pl['splits_b'] * (M - pl['m_rem0']) * Co if pl['splits_b'] > 1 else 0)
            assert partial == pl['partial'] and partial * 4 <= need
            planned += partial > 0
    assert planned >= 20            # the mid-size layers do get balanced
    # the l3 bottleneck 1x1 (16800 x 256, K = 1024) on 64x128 tiles: 526 tiles = 512 whole + 14 cut into small ranges
    pl = _x3p_plan(3, 16800, 256, 1024, 1)
    assert pl['tiles_a'] == 512 and pl['splits_a'] == 1 and pl['splits_b'] >= 8


def test_x3p_kernels_never_move_a_register_with_a_load_in_flight():
    """tools/x3p_check_isa.py on the compiled conv_x3.hip: the activation loads of the K loop are inline asm whose results
    are outstanding across barriers and the loop's back edge; a compiler-placed copy of such a register would read stale
    data.  Static check of all 24 instantiations (16 on the bf16 form, 8 on the H2 arithmetic; no GPU), plus: no scratch, no spills."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'x3p_check_isa.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert '24 conv_x3p_kernel instantiations checked, 0 findings' in r.stdout


def test_parameter_order_is_the_reference_registration_order():
    """An optimizer state dict numbers parameters by position in `model.parameters()` (mmcv hands that iterator to
    torch.optim.SGD, apis/train.py:86): a reference checkpoint's momentum buffers land on the right tensors only if this
    package registers its modules in the reference's order.  golden_util.reference_parameter_order derives that order
    from the reference's sources; R50 and R101(-DCN) must match it name by name."""
    from golden_util import reference_parameter_order
    from htd_amd.configs import build_htd_detector
    from oracle import detector as D
    for depth, dcn in ((50, False), (101, False), (101, True)):
        model = build_htd_detector(depth, dcn=dcn)
        assert [n for n, _ in model.named_parameters()] == reference_parameter_order(D.state_shapes(depth, dcn)), (depth, dcn)
