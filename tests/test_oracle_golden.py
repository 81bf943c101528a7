"""The CPU oracle (oracle/) against fixtures produced by the reference's own files
(tests/golden/make_golden.py) and against the reference tests' known answers."""
import numpy as np
import pytest
import torch

from golden_util import demo_inputs, digest, seeded_state_dict, seeded_tensor
from oracle import boxes as B
from oracle import detector as D
from oracle import ops


def T(a):
    return torch.from_numpy(np.asarray(a))


def check_digest(g, key, t, rtol=2e-4, atol=1e-5):
    sums, sample = digest(t)
    ref = g[key + '.sample']
    np.testing.assert_allclose(sample, ref, rtol=rtol, atol=atol * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(sums[1], g[key + '.sums'][1], rtol=1e-4)


def test_anchors_exact(golden):
    g = golden('anchors')
    sizes = [tuple(s) for s in g['sizes']]
    strides = [4, 8, 16, 32, 64]
    anchors = B.grid_anchors(sizes, strides)
    flags = B.valid_flags(sizes, strides, tuple(g['pad_shape']))
    for i in range(5):
        assert torch.equal(anchors[i], T(g[f'anchors{i}']))
        assert torch.equal(flags[i], T(g[f'flags{i}']))
        assert torch.equal(B.gen_base_anchors(strides[i], [8], [0.5, 1.0, 2.0]), T(g[f'base{i}']))
    # reference KAT tests/test_anchor.py:22-40
    kat = B.grid_anchors([(2, 2)], [10], scales=[1.], ratios=[1.], base_sizes=[10])[0]
    assert torch.equal(kat, torch.tensor([[-5., -5., 5., 5.], [5., -5., 15., 5.], [-5., 5., 5., 15.], [5., 5., 15., 15.]]))
    assert torch.equal(kat, T(g['kat_anchor']))
    kat2 = B.grid_anchors([(2, 2)], [(10, 20)], scales=[1.], ratios=[1.], base_sizes=[10])[0]
    assert torch.equal(kat2, T(g['kat_anchor_xy']))


def test_box_math(golden):
    g = golden('box_math')
    b1, b2 = T(g['b1']), T(g['b2'])
    assert torch.equal(B.bbox_overlaps(b1, b2), T(g['iou']))
    assert torch.equal(B.bbox_overlaps(b1, b2, mode='iof'), T(g['iof']))
    assert torch.equal(B.bbox_overlaps(b1[:9], b2, is_aligned=True), T(g['iou_aligned']))
    torch.testing.assert_close(B.bbox2delta(b1[:9], b2, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)), T(g['deltas']),
                               rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(B.delta2bbox(b1, T(g['rnd']), (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2),
                                            max_shape=(80, 90, 3)), T(g['dec']), rtol=1e-6, atol=1e-5)
    torch.testing.assert_close(B.delta2bbox(b1, T(g['rnd']) * 10), T(g['dec_noclip']), rtol=1e-6, atol=1e-4)
    # reference docstring KAT delta_xywh_bbox_coder.py:156-169
    dec = B.delta2bbox(T(g['kat_rois']), T(g['kat_deltas']), max_shape=(32, 32))
    expected = torch.tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                             [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]])
    torch.testing.assert_close(dec, expected, rtol=0, atol=1e-4)
    torch.testing.assert_close(dec, T(g['kat_dec']), rtol=1e-6, atol=1e-6)


def test_assigner(golden):
    g = golden('assigner')
    r = B.max_iou_assign(T(g['kat_bboxes']), T(g['kat_gts']), torch.LongTensor([2, 3]), 0.5, 0.5)
    assert r.gt_inds.tolist() == [1, 0, 2, 0]          # tests/test_assigner.py:14-35
    assert torch.equal(r.gt_inds, T(g['kat_gt_inds'])) and torch.equal(r.labels, T(g['kat_labels']))
    for tag, kw in [('rpn', dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)),
                    ('rcnn', dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False))]:
        r = B.max_iou_assign(T(g[f'{tag}_bboxes']), T(g[f'{tag}_gts']), T(g[f'{tag}_gt_labels']), **kw)
        assert torch.equal(r.gt_inds, T(g[f'{tag}_gt_inds']))
        assert torch.equal(r.labels, T(g[f'{tag}_labels']))
        assert torch.equal(r.max_overlaps, T(g[f'{tag}_max_overlaps']))
    r = B.max_iou_assign(T(g['kat_bboxes']), torch.empty(0, 4), None, 0.5, 0.5)   # tests/test_assigner.py:110-130
    assert torch.equal(r.gt_inds, T(g['empty_gt_inds'])) and (r.gt_inds == 0).all()
    r = B.max_iou_assign(torch.empty(0, 4), T(g['kat_gts']), None, 0.5, 0.5)
    assert r.gt_inds.numel() == 0


def test_sampler_replays_cpu_rng(golden):
    g = golden('sampler')
    pb, gb, gl = T(g['bboxes']), T(g['gts']), T(g['gt_labels'])
    ar = B.max_iou_assign(pb, gb, gl, 0.3, 0.3, 0.3, match_low_quality=False)
    torch.manual_seed(int(g['seed']))
    s = B.random_sample(ar, pb, gb, gl, num=64, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    for k in ('pos_inds', 'neg_inds', 'pos_is_gt', 'pos_assigned_gt_inds', 'pos_gt_labels', 'pos_bboxes', 'neg_bboxes'):
        assert torch.equal(getattr(s, k), T(g[k])), k


def test_sfa(golden):
    g = golden('sfa')
    sd = seeded_state_dict({k: v for k, v in D.state_shapes().items() if k.startswith('roi_head.glbctx_head.')})
    sd = {k.replace('roi_head.glbctx_head.', ''): v for k, v in sd.items()}
    from golden_util import seeded_state_value
    sd = {k: T(seeded_state_value('sfa.' + k, v.shape)).requires_grad_() for k, v in sd.items()}
    p6 = seeded_tensor('sfa.p6', (2, 256, 5, 7)).requires_grad_()
    mc, gf = D.sfa_forward(sd, [p6], prefix='')
    loss = D.sfa_loss(mc, [T(g['labels0']), T(g['labels1'])], 3.0)
    loss.backward()
    torch.testing.assert_close(mc, T(g['mc_pred']), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gf, T(g['global_feat']), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loss, T(g['loss']), rtol=1e-6, atol=1e-6)
    check_digest(g, 'grad_p6', p6.grad)
    check_digest(g, 'grad_fc_w', sd['fc.weight'].grad)
    check_digest(g, 'grad_conv0_w', sd['convs.0.conv.weight'].grad)


def _head_sd(prefix_key, seed_prefix):
    from golden_util import seeded_state_value
    return {k: T(seeded_state_value(seed_prefix + k[len(prefix_key):], s)).requires_grad_()
            for k, s in D.state_shapes().items() if k.startswith(prefix_key)}


def test_stage1_head(golden):
    g = golden('stage1_head')
    sd = _head_sd('roi_head.bbox_head.0.', 'head0.')
    x = seeded_tensor('head0.x', (24, 256, 7, 7)).requires_grad_()
    cls, reg = D.shared2fc_forward(sd, x)
    torch.testing.assert_close(cls, T(g['cls']), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(reg, T(g['reg']), rtol=1e-5, atol=1e-5)
    losses = D.bbox_loss(cls, reg, T(g['labels']), T(g['label_weights']), T(g['bbox_targets']), T(g['bbox_weights']))
    for k in ('loss_cls', 'loss_bbox', 'acc'):
        torch.testing.assert_close(losses[k].reshape(-1), T(g[k]).reshape(-1), rtol=1e-5, atol=1e-6)
    (losses['loss_cls'] + losses['loss_bbox']).backward()
    check_digest(g, 'grad_x', x.grad)
    check_digest(g, 'grad_fc_cls_w', sd['roi_head.bbox_head.0.fc_cls.weight'].grad)


def test_pgraph(golden):
    g = golden('pgraph')
    sd = _head_sd('roi_head.bbox_head.1.', 'head1.')
    sd.update(_head_sd('roi_head.bbox_head.0.', 'head0.'))
    rois, pos_idx = T(g['rois']), T(g['pos_idx'])
    assert torch.equal(B.map_roi_levels(rois, 4), T(g['target_lvls']))
    x_cls = seeded_tensor('head1.x_cls', (40, 256, 7, 7)).requires_grad_()
    enhanced = seeded_tensor('head1.enhanced', (10, 256, 7, 7)).requires_grad_()
    gfeat = seeded_tensor('head1.gfeat', (2, 256, 1, 1)).requires_grad_()
    cls, reg = D.htd_bbox_head_forward(sd, x_cls, x_cls[pos_idx], rois, enhanced, rois[pos_idx], gfeat)
    torch.testing.assert_close(cls, T(g['cls']), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(reg, T(g['reg']), rtol=1e-4, atol=1e-4)
    ((cls * seeded_tensor('head1.gcls', cls.shape)).sum() + (reg * seeded_tensor('head1.greg', reg.shape)).sum()).backward()
    h = 'roi_head.bbox_head.1.'
    for key, t in [('grad_x_cls', x_cls.grad), ('grad_enhanced', enhanced.grad), ('grad_global', gfeat.grad),
                   ('grad_fc_cls_0_w', sd['roi_head.bbox_head.0.fc_cls.weight'].grad),
                   ('grad_fc_cls_0_b', sd['roi_head.bbox_head.0.fc_cls.bias'].grad),
                   ('grad_graph0_w', sd[h + 'graph_lvl0_cls.weight'].grad),
                   ('grad_graph3_w', sd[h + 'graph_lvl3_cls.weight'].grad),
                   ('grad_fcs0_w', sd[h + 'fcs.0.weight'].grad), ('grad_conv0_w', sd[h + 'convs.0.conv.weight'].grad),
                   ('grad_gn0_w', sd[h + 'convs.0.gn.weight'].grad), ('grad_fc_reg_w', sd[h + 'fc_reg.weight'].grad)]:
        check_digest(g, key, t, rtol=2e-3, atol=2e-5)


def test_ba(golden):
    g = golden('ba')
    sd = _head_sd('roi_head.bbox_roi_extractor.1.', 'ba.')
    feats = [seeded_tensor(f'ba.feat{i}', (2, 256, 64 // s, 96 // s)).requires_grad_() for i, s in enumerate((1, 2, 4, 8))]
    out = D.ba_extract(sd, feats, T(g['rois']))
    torch.testing.assert_close(out, T(g['out']), rtol=1e-5, atol=1e-5)
    (out * seeded_tensor('ba.go', out.shape)).sum().backward()
    for i, f in enumerate(feats):
        check_digest(g, f'grad_feat{i}', f.grad)
    check_digest(g, 'grad_conv1_w', sd['roi_head.bbox_roi_extractor.1.conv1.weight'].grad)
    check_digest(g, 'grad_conv2_w', sd['roi_head.bbox_roi_extractor.1.conv2.weight'].grad)
    # n == 1 (the reference's .squeeze() asserts there): defined as the n-row of a 2-row call
    one = D.ba_extract(sd, feats, T(g['rois'])[1:2])
    torch.testing.assert_close(one, out[1:2].detach(), rtol=1e-5, atol=1e-5)


def small_cfg():
    cfg = D.htd_config(50)
    cfg['train_cfg']['rpn_proposal'].update(nms_pre=200, nms_post=100, max_num=100)
    for r in cfg['train_cfg']['rcnn']:
        r['sampler']['num'] = 48
    cfg['test_cfg']['rpn'].update(nms_pre=100, nms_post=60, max_num=60)
    cfg['test_cfg']['rcnn']['score_thr'] = 0.001
    return cfg


def detector_inputs(g):
    H, W = int(g['H']), int(g['W'])
    imgs, gts, labels = demo_inputs(2, H, W, np.random.RandomState(0))
    imgs = (imgs - 0.5) * 4
    iw = int(g['img_w'])
    metas = [dict(img_shape=(H, iw, 3), pad_shape=(H, W, 3), ori_shape=(H, iw, 3),
                  scale_factor=np.array([1, 1, 1, 1], dtype=np.float32), flip=False) for _ in range(2)]
    gts = [np.minimum(x, np.array([iw, H, iw, H], dtype=np.float32)) for x in gts]
    return T(imgs), metas, [T(x) for x in gts], [T(x) for x in labels]


def test_detector_train_and_test(golden):
    """Whole path: losses, gradient digests, proposals and detections of the reference run."""
    g = golden('detector')
    cfg = small_cfg()
    sd = {k: v.requires_grad_(v.dtype.is_floating_point and 'running' not in k)
          for k, v in seeded_state_dict(D.state_shapes(50), prefix='det.').items()}
    img, metas, gts, labels = detector_inputs(g)
    for i in range(2):
        np.testing.assert_array_equal(gts[i].numpy(), g[f'gt{i}'])
        np.testing.assert_array_equal(labels[i].numpy(), g[f'label{i}'])
    torch.manual_seed(int(g['seed_sampler']))
    losses = D.forward_train(sd, img, metas, gts, labels, cfg)
    loss, log_vars = D.parse_losses(losses)
    for k, v in log_vars.items():
        np.testing.assert_allclose(v, float(g['loss.' + k]), rtol=2e-4, atol=1e-5, err_msg=k)
    loss.backward()
    for k in [f[5:-5] for f in g.files if f.startswith('grad.') and f.endswith('.sums')]:
        gr = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        check_digest(g, 'grad.' + k, gr, rtol=5e-3, atol=5e-4)
    with torch.no_grad():
        props, dets = D.simple_test(sd, img, metas, cfg)
    for i in range(2):
        torch.testing.assert_close(props[i], T(g[f'test_props{i}']), rtol=1e-4, atol=1e-3)
        d, l = dets[i]
        ref = g[f'test_dets{i}']
        # the reference returns per-class arrays (bbox2result): compare as class-sorted sets
        mine = torch.cat([d, l[:, None].float()], 1).numpy()
        mine = mine[np.lexsort((-mine[:, 4], mine[:, 5]))]
        refs = ref[np.lexsort((-ref[:, 4], ref[:, 5]))]
        assert mine.shape == refs.shape
        np.testing.assert_array_equal(mine[:, 5], refs[:, 5])
        np.testing.assert_allclose(mine[:, :5], refs[:, :5], rtol=1e-3, atol=1e-2)


def test_detector_index_trail_and_stage_logits(golden):
    """north_star's parity clause at path level, oracle side: fed the reference run's RPN logits, the proposal stage
    keeps the SAME candidates in the SAME order (torch.equal on the keep rows and on the anchor identity of every
    proposal, rpn_head.py:122-168); fed its stage inputs, both RoI stages answer with the reference's cls / box logits
    within 1e-4 -- in the test path and in the training path (sampler replayed)."""
    g = golden('detector')
    cfg = small_cfg()
    sd = {k: v.requires_grad_(v.dtype.is_floating_point and 'running' not in k)
          for k, v in seeded_state_dict(D.state_shapes(50), prefix='det.').items()}
    img, metas, gts, labels = detector_inputs(g)
    cls = [T(g[f'rpn_cls{l}']) for l in range(5)]
    reg = [T(g[f'rpn_reg{l}']) for l in range(5)]
    trace = []
    props = D.rpn_get_bboxes(cls, reg, metas, cfg['test_cfg']['rpn'], cfg, cfg['strides'], trace=trace)
    for i in range(2):
        keep, anchor_ids = trace[i][:2]
        assert torch.equal(keep, T(g[f'test_keep{i}']))
        assert torch.equal(anchor_ids, T(g[f'test_prop_anchor{i}']))
        assert torch.equal(props[i], T(g[f'test_props{i}']))                  # same arithmetic on the same logits: exact
    with torch.no_grad():
        x = D.extract_feat(sd, img, cfg)
        own_cls, own_reg = D.rpn_forward(sd, x)
        for l in range(5):                                                    # the oracle's own RPN logits
            torch.testing.assert_close(own_cls[l], cls[l], rtol=0, atol=1e-4)
            torch.testing.assert_close(own_reg[l], reg[l], rtol=0, atol=1e-4)
        tr = {}
        D.roi_head_simple_test(sd, x, [T(g[f'test_props{i}']) for i in range(2)], metas, cfg, trace=tr)
    for st in (0, 1):
        torch.testing.assert_close(tr[f'rois{st}'], T(g[f'test_s{st}_rois']), rtol=0, atol=1e-3)
        torch.testing.assert_close(tr[f'cls{st}'], T(g[f'test_s{st}_cls']), rtol=0, atol=1e-4)
        torch.testing.assert_close(tr[f'reg{st}'], T(g[f'test_s{st}_reg']), rtol=0, atol=1e-4)
    torch.manual_seed(int(g['seed_sampler']))
    tr = {}
    D.forward_train(sd, img, metas, gts, labels, cfg, tr)
    for st in (0, 1):
        torch.testing.assert_close(tr[f'rois{st}'], T(g[f'train_s{st}_rois']), rtol=0, atol=1e-3)
        torch.testing.assert_close(tr[f'cls{st}'].detach(), T(g[f'train_s{st}_cls']), rtol=0, atol=1e-4)
        torch.testing.assert_close(tr[f'reg{st}'].detach(), T(g[f'train_s{st}_reg']), rtol=0, atol=1e-4)


def test_aug_test_matches_reference_fixture(golden):
    """Test-time augmentation (two_stage.py:213-222, rpn_test_mixin.py:39-59, htd_roi_head.py:388-433, merge_augs.py):
    the oracle's restatement against the outputs of the reference's own aug_test (tests/golden/aug_test.npz)."""
    from golden_util import aug_inputs, match_detections
    g = golden('aug_test')
    cfg = small_cfg()
    sd = seeded_state_dict(D.state_shapes(50), prefix='det.')
    imgs, metas = aug_inputs()
    with torch.no_grad():
        props, (dets, labels) = D.aug_test(sd, [T(i) for i in imgs], metas, cfg)
    assert props.shape == g['proposals'].shape
    np.testing.assert_allclose(props.numpy(), g['proposals'], rtol=1e-4, atol=1e-3)
    match_detections(torch.cat([dets, labels[:, None].float()], 1).numpy(), g['dets'])


def test_resnext_stage_matches_reference_fixture(golden):
    """ResNeXt bottlenecks (backbones/resnext.py:9-84; groups = 8, base_width = 4, stride-2 stage of two blocks): the
    oracle's bottleneck(groups=...) against the reference's own module outputs and gradients."""
    g = golden('resnext_stage')
    shapes = {}
    w = 32
    for b, cin in ((0, 64), (1, 256)):
        shapes.update({f'{b}.conv1.weight': (w, cin, 1, 1), f'{b}.conv2.weight': (w, 4, 3, 3),
                       f'{b}.conv3.weight': (256, w, 1, 1)})
        for n, c in (('bn1', w), ('bn2', w), ('bn3', 256)):
            for k in ('weight', 'bias', 'running_mean', 'running_var'):
                shapes[f'{b}.{n}.{k}'] = (c,)
    shapes['0.downsample.0.weight'] = (256, 64, 1, 1)
    for k in ('weight', 'bias', 'running_mean', 'running_var'):
        shapes[f'0.downsample.1.{k}'] = (256,)
    assert tuple(g['conv2_shape']) == shapes['0.conv2.weight']
    sd = {'blk' + k: v.requires_grad_('running' not in k)
          for k, v in seeded_state_dict(shapes, prefix='xblk.').items()}
    x = torch.randn(2, 64, 12, 14, generator=torch.Generator().manual_seed(1)).requires_grad_()
    y = D.bottleneck(sd, 'blk1', D.bottleneck(sd, 'blk0', x, 2, groups=8), 1, groups=8)
    torch.testing.assert_close(y.detach(), T(g['y']), rtol=1e-4, atol=1e-5)
    y.backward(torch.randn(y.shape, generator=torch.Generator().manual_seed(2)))
    check_digest(g, 'gx', x.grad)
    for k in [f[5:-5] for f in g.files if f.startswith('grad.') and f.endswith('.sums')]:
        check_digest(g, 'grad.' + k, sd['blk' + k].grad)
