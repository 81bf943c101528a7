"""BASELINE.json configs[2], [3], [4] on a real MI355X, each against the CPU oracle at a size the oracle finishes:

  configs[2]  HTD-R101 bf16 train step          (bf16 trunk / FC stacks, fp32 master weights, fp32 RoI ops and losses)
  configs[3]  HTD-R101-DCN bf16 train step      (+ 30 deformable 3x3 layers on the bf16 kernels)
  configs[4]  HTD-R101 inference, 512 proposals per image into the RoI head, hard NMS

bf16 tolerance (stated, not a smoke bound).  A bf16 value carries 8 significand bits: rounding an activation or a
weight perturbs it by at most 2^-9 relative (RMS 2^-9 / sqrt(3) ~ 1.1e-3).  The trunk rounds ~3 activations and ~3
weight tensors per bottleneck over 33 bottlenecks plus the FPN, independent perturbations that add in quadrature:
relative L2 error of a pyramid level ~ 1.1e-3 * sqrt(2 * 100) ~ 1.6e-2 worst case.  The bounds below are 2x what this
model predicts and are checked as RELATIVE L2 ERRORS against the fp32 oracle run on the same fp32 master weights:
    pyramid levels            <= 3e-2       losses (each)        <= 3e-2 relative (+1e-3 abs)
    RPN logits                <= 3e-2       gradients (rel. L2)  <= 1.5e-1 trunk / 6e-2 heads / 2e-1 PGraph-coupled and BA / SFA
(the model's 1.6e-2 is met ON AVERAGE: over 8 (image, weight) seed pairs the worst pyramid level is 1.98e-2 -- the round-2
bounds of 2e-2 / 1e-1 sat on the worst case of a sweep they had not run; see test_bf16_gradient_bounds_hold_over_seeds)
(a trunk weight gradient is the product of a forward activation and a back-propagated gradient that has itself been
rounded to bf16 at every layer on the way down, plus the ReLU masks that flip where a pre-activation sits within a
rounding error of zero: about sqrt(2) x the forward error from each factor, measured 5e-2 .. 7.5e-2 on R101)
configs[3] (R101-DCN): a deformable layer samples its input at p + offset and the offsets come out of a convolution of
the bf16 activations: an offset error of d pixels moves the bilinear sample by d * |grad x|, and the gradient with
respect to the offsets is a finite difference of neighbouring pixels.  On the rough feature maps of a seeded (untrained)
network with pixel-sized offsets that path is chaotic (a 3 % forward difference de-correlates d x / d p: trunk gradients
50 % apart while every loss agrees to 0.5 %), which says nothing about the kernels.  The bf16 comparison therefore uses
early-training offsets (seeded conv_offset x 0.1; the reference initialises it to zero, resnet.py:608-612) and the bounds
    pyramid / RPN logits <= 3e-2,  losses <= 3e-2,  trunk gradients <= 2e-1,  heads <= 6e-2 (PGraph-coupled, BA / SFA <= 2e-1);
the fp32 R101-DCN step with FULL-size offsets is held to the oracle at fp32 bounds in
test_r101_fp32_train_step_against_the_oracle[dcn], and single deformable layers to 1e-4 in tests/test_gpu_dcn.py.
BA and SFA convolutions see only a handful of pixels at this image size (P6 is 2x3): their weight gradients are sums of
few terms and carry the pyramid's error almost unaveraged, hence the wider bound.
The PGraph-coupled group (`fcs.0`, `graph_lvl*`, measured up to 1.4e-1): round 3 put it down to "soft-max amplification, not
measured".  Measured in round 4 (tools/pgraph_sensitivity.py, profiles/r04_pgraph_sensitivity.log: the branch in pure fp32 on
RoI tiles x and x (1 + e), e ~ N(0, s^2)): with s = 1 % the weight gradients move by 7.9e-2 (`fcs.0`), 5.9e-2 (`fcs.2`),
2.6e-2 / 5.5e-2 (`graph_lvl0 / 1`) and 5e-3 (`fc_cls`) relative L2 -- and those of the PLAIN stage-1 stack by 1.2e-1 / 8.4e-2
(`shared_fcs.0 / 1`).  So the graph layers amplify a perturbation 2.6-5.5x, LESS than a ReLU FC stack of the untrained
network does (8-12x: pre-activations of seeded weights crowd around zero and the masks flip); the 14 % of `fcs.0` is the FC
stack's own sensitivity to a pyramid that is 1-2 % off (it runs on the plain AND the fused tiles: twice the flips), not the
soft-max.  The bound of the group stays where the 8-seed sweep put it.
"""
import numpy as np
import pytest
import torch

from golden_util import demo_inputs, load_seeded_, seeded_state_dict
from test_gpu_detector import ReplaySampler, T

pytestmark = pytest.mark.gpu


OFFSET_SCALE = 0.1


def rel_l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _small(cfg_obj, ocfg):
    cfg_obj.train_cfg.rpn_proposal.update(nms_pre=200, nms_post=100, max_num=100)
    ocfg['train_cfg']['rpn_proposal'].update(nms_pre=200, nms_post=100, max_num=100)
    for r in cfg_obj.train_cfg.rcnn:
        r.sampler.num = 48
    for r in ocfg['train_cfg']['rcnn']:
        r['sampler']['num'] = 48


def bf16_step_errors(dcn, data_seed=11, weight_seed=1234):
    """One HTD-R101(-DCN) bf16 train step on the GPU against the fp32 oracle on the same master weights, samples replayed.
    -> (feature / logit errors, {loss: (product, oracle)}, {group: {parameter: relative L2 error of its gradient}})."""
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.core import set_randperm
    from oracle import detector as D
    dev = torch.device('cuda:0')
    H, W, B = 128, 160, 2
    imgs, gts, labels = demo_inputs(B, H, W, np.random.RandomState(data_seed))
    imgs = (imgs - 0.5) * 4
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3),
                  scale_factor=np.array([1, 1, 1, 1], dtype=np.float32), flip=False) for _ in range(B)]
    ocfg = D.htd_config(101, dcn)
    cfg = htd_config(101, dcn=dcn)
    _small(cfg, ocfg)
    shapes = D.state_shapes(101, dcn)
    sd = {k: v.requires_grad_(v.dtype.is_floating_point and 'running' not in k)
          for k, v in seeded_state_dict(shapes, prefix='det.', seed=weight_seed).items()}
    if dcn:                                  # early-training offsets (the reference initialises conv_offset to zero)
        with torch.no_grad():
            for k, v in sd.items():
                if 'conv_offset' in k:
                    v.mul_(OFFSET_SCALE)
    torch.manual_seed(21)
    trace = {}
    ref_losses = D.forward_train(sd, T(imgs), metas, [T(x) for x in gts], [T(x) for x in labels], ocfg, trace)
    ref_loss, ref_log = D.parse_losses(ref_losses)
    ref_loss.backward()

    det = build_htd_detector(cfg=cfg, bf16=True)
    load_seeded_(det, 'det.', seed=weight_seed)
    if dcn:
        with torch.no_grad():
            for k, p in det.named_parameters():
                if 'conv_offset' in k:
                    p.mul_(OFFSET_SCALE)
    det = det.to(dev).train()
    assert det.backbone.compute_dtype == torch.bfloat16
    set_randperm(lambda n, device: torch.randperm(n).to(device))
    try:
        torch.manual_seed(21)
        gts_d, labels_d = [T(x).to(dev) for x in gts], [T(x).to(dev) for x in labels]
        x = det.extract_feat(T(imgs).to(dev))
        errs = {f'P{i + 2}': rel_l2(a.float(), b) for i, (a, b) in enumerate(zip(x, trace['feats']))}
        cls, reg = det.rpn_head(x)
        errs['rpn_cls'] = max(rel_l2(a, b) for a, b in zip(cls, trace['rpn_cls']))
        errs['rpn_reg'] = max(rel_l2(a, b) for a, b in zip(reg, trace['rpn_reg']))
        # RPN losses end to end; RoI head on the oracle's proposals and sample picks (an IoU that sits on a threshold
        # flips with any perturbation and reshuffles the random permutation: that is sampling, not arithmetic)
        losses, _ = det.rpn_head.forward_train(x, metas, gts_d, proposal_cfg=det.train_cfg.rpn_proposal)
        head = det.roi_head
        saved = list(head.bbox_sampler)
        try:
            head.bbox_sampler = [ReplaySampler(saved[i], trace['samples'][i]) for i in range(2)]
            x32 = tuple(f.float() for f in x)      # the RoI head reads the pyramid in fp32 (force_fp32 sites), as two_stage does
            losses.update(head.forward_train(x32, metas, [p.to(dev) for p in trace['proposals']], gts_d, labels_d))
        finally:
            head.bbox_sampler = saved
    finally:
        set_randperm(None)
    loss, log_vars = det._parse_losses(losses)
    det.zero_grad()
    loss.backward()
    losses_out = {k: (v, ref_log[k]) for k, v in log_vars.items() if 'acc' not in k}
    params = dict(det.named_parameters())
    groups = dict(trunk=['backbone.layer2.0.conv1.weight', 'backbone.layer3.10.conv2.weight', 'backbone.layer4.2.conv3.weight',
                         'neck.lateral_convs.2.conv.weight', 'neck.fpn_convs.0.conv.weight', 'rpn_head.rpn_conv.weight'] +
                  (['backbone.layer3.5.conv2.conv_offset.weight'] if dcn else []),
                  heads=['roi_head.bbox_head.0.shared_fcs.1.weight', 'roi_head.bbox_head.0.fc_cls.weight',
                         'roi_head.bbox_head.1.convs.1.conv.weight'],
                  graph=['roi_head.bbox_head.1.fcs.0.weight', 'roi_head.bbox_head.1.graph_lvl0_cls.weight'],
                  small=['roi_head.bbox_roi_extractor.1.conv1.weight', 'roi_head.glbctx_head.convs.0.conv.weight'])
    grads = {}
    for g, names in groups.items():
        grads[g] = {}
        for k in names:
            a = params[k].grad.detach().cpu() if params[k].grad is not None else torch.zeros_like(params[k]).cpu()
            b = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
            assert params[k].grad.dtype == torch.float32                 # fp32 master gradients
            grads[g][k] = rel_l2(a.reshape(b.shape), b)
    return errs, losses_out, grads


def bf16_bounds(dcn):
    """Bounds of the bf16 step (relative L2 against the fp32 oracle), per group; the module docstring has the error model,
    test_bf16_gradient_bounds_hold_over_seeds the measured worst cases they are set from."""
    return dict(feat=3e-2, loss=3e-2, trunk=2e-1 if dcn else 1.5e-1, heads=6e-2, graph=2e-1, small=2e-1)


def check_bf16_step(dcn, errs, losses, grads):
    b = bf16_bounds(dcn)
    failures = []
    print('\nbf16 vs fp32 oracle (dcn=%s): relative L2 errors' % dcn, {k: round(v, 5) for k, v in errs.items()})
    failures += [(k, v, b['feat']) for k, v in errs.items() if v > b['feat']]
    for k, (v, r) in losses.items():
        print('  loss %-14s %.5f  oracle %.5f  rel %.2e' % (k, v, r, abs(v - r) / max(abs(r), 1e-9)))
        if abs(v - r) > b['loss'] * abs(r) + 1e-3:
            failures.append((k, v, r))
    for g in ('trunk', 'heads', 'graph', 'small'):
        for k, e in grads[g].items():
            print('  grad %-48s rel L2 %.2e' % (k, e))
            if e > b[g]:
                failures.append((k, e, b[g]))
    return failures


@pytest.mark.parametrize('dcn', [False, True], ids=['configs2_r101_bf16', 'configs3_r101_dcn_bf16'])
def test_r101_bf16_train_step_against_the_oracle(dcn):
    failures = check_bf16_step(dcn, *bf16_step_errors(dcn))
    assert not failures, failures


@pytest.mark.parametrize('dcn', [False, True], ids=['configs2_r101_bf16', 'configs3_r101_dcn_bf16'])
def test_bf16_gradient_bounds_hold_over_seeds(dcn):
    """the bounds of the bf16 step must not be fitted to one draw.  The same comparison over other images and
    other seeded weights (2 further (data, weight) seed pairs in the suite; HTD_BF16_SEEDS=n extends the sweep) has to stay
    inside the SAME bounds; the largest error per group is printed next to its bound.
    Measured on MI355X over 8 pairs per configuration (`HTD_BF16_SEEDS=8`, profiles/r03_bf16_bounds.log), largest relative
    L2 error, with the bound in brackets:
                      features  losses    trunk grads     stage-1 head + reg convs   stage-2 fcs.0 / graph   BA / SFA
        configs[2]    1.98e-2   1.90e-2   1.01e-1 [1.5e-1]   4.05e-2 [6e-2]           1.39e-1 [2e-1]          1.42e-1 [2e-1]
        configs[3]    1.88e-2   1.52e-2   1.48e-1 [2e-1]     4.46e-2 [6e-2]           8.04e-2 [2e-1]          1.35e-1 [2e-1]
                      [3e-2]    [3e-2]
    Round 2's single draw had hidden two things the sweep shows: (1) the 2e-2 feature / loss bounds and the 1e-1 trunk bound
    sat ON the worst case (1.98e-2, 1.90e-2, 1.01e-1), and (2) the gradients of the stage-2 classification branch that
    passes through PGraph (fcs.0, graph_lvl*_cls) spread far more than the other head gradients (1.3e-2 .. 1.4e-1 across
    draws, the rest <= 4.5e-2): A_glob = softmax((1 - M) * sim) exponentiates a similarity of bf16-rounded FC features, so
    a 2^-9 relative rounding of a logit of magnitude s becomes a relative error of about s * 2^-9 in an attention weight,
    and the logits of a seeded (untrained) network are not small (the amplification itself was not measured separately;
    the same branch in fp32 agrees with the oracle to 1e-4, test_r101_fp32_train_step_against_the_oracle).  That group now
    has its own bound.  Every bound is 1.35 - 1.5x the worst of the 8 draws of its configuration."""
    import os
    n = int(os.environ.get('HTD_BF16_SEEDS', '2'))
    worst = dict(trunk=0.0, heads=0.0, graph=0.0, small=0.0, feat=0.0, loss=0.0)
    failures = []
    for i in range(n):
        data_seed, weight_seed = 101 + 7 * i, 4321 + 13 * i
        errs, losses, grads = bf16_step_errors(dcn, data_seed, weight_seed)
        failures += [(data_seed, weight_seed) + f for f in check_bf16_step(dcn, errs, losses, grads)]
        for g in ('trunk', 'heads', 'graph', 'small'):
            worst[g] = max(worst[g], max(grads[g].values()))
        worst['feat'] = max(worst['feat'], max(errs.values()))
        worst['loss'] = max(worst['loss'], max(abs(v - r) / max(abs(r), 1e-9) for v, r in losses.values()))
    b = bf16_bounds(dcn)
    print('\nWORST over %d seed pairs (dcn=%s): ' % (n, dcn) +
          '  '.join('%s %.2e (bound %.1e)' % (g, worst[g], b[g]) for g in ('feat', 'loss', 'trunk', 'heads', 'graph', 'small')))
    assert not failures, failures


@pytest.mark.parametrize('dcn', [False, True], ids=['r101', 'dcn'])
def test_r101_fp32_train_step_against_the_oracle(dcn):
    """The same R101(-DCN) step in fp32, deformable offsets at full seeded size (about a pixel): losses at the bounds of
    the R50 fixtures, gradients -- through all 30 deformable layers and their offset branches -- within 1e-2 relative L2
    (the offset path multiplies rounding differences; the same comparison without DCN gives 1e-4 .. 1e-3)."""
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.core import set_randperm
    from oracle import detector as D
    dev = torch.device('cuda:0')
    H, W, B = 128, 160, 2
    imgs, gts, labels = demo_inputs(B, H, W, np.random.RandomState(11))
    imgs = (imgs - 0.5) * 4
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3),
                  scale_factor=np.array([1, 1, 1, 1], dtype=np.float32), flip=False) for _ in range(B)]
    ocfg, cfg = D.htd_config(101, dcn), htd_config(101, dcn=dcn)
    _small(cfg, ocfg)
    sd = {k: v.requires_grad_(v.dtype.is_floating_point and 'running' not in k)
          for k, v in seeded_state_dict(D.state_shapes(101, dcn), prefix='det.').items()}
    torch.manual_seed(21)
    trace = {}
    ref_loss, ref_log = D.parse_losses(D.forward_train(sd, T(imgs), metas, [T(x) for x in gts], [T(x) for x in labels], ocfg, trace))
    ref_loss.backward()
    det = load_seeded_(build_htd_detector(cfg=cfg), 'det.').to(dev).train()
    set_randperm(lambda n, device: torch.randperm(n).to(device))
    try:
        torch.manual_seed(21)
        gts_d, labels_d = [T(x).to(dev) for x in gts], [T(x).to(dev) for x in labels]
        x = det.extract_feat(T(imgs).to(dev))
        for i, (a, b) in enumerate(zip(x, trace['feats'])):
            assert rel_l2(a, b) <= 1e-5, (i, rel_l2(a, b))
        losses, _ = det.rpn_head.forward_train(x, metas, gts_d, proposal_cfg=det.train_cfg.rpn_proposal)
        head = det.roi_head
        saved = list(head.bbox_sampler)
        try:
            head.bbox_sampler = [ReplaySampler(saved[i], trace['samples'][i]) for i in range(2)]
            losses.update(head.forward_train(x, metas, [p.to(dev) for p in trace['proposals']], gts_d, labels_d))
        finally:
            head.bbox_sampler = saved
    finally:
        set_randperm(None)
    loss, log_vars = det._parse_losses(losses)
    for k, v in log_vars.items():
        np.testing.assert_allclose(v, ref_log[k], rtol=5e-4, atol=1e-4, err_msg=k)
    det.zero_grad()
    loss.backward()
    params = dict(det.named_parameters())
    names = ['backbone.layer2.0.conv1.weight', 'backbone.layer3.10.conv2.weight', 'backbone.layer4.2.conv3.weight',
             'neck.fpn_convs.0.conv.weight', 'roi_head.bbox_head.1.fcs.0.weight']
    if dcn:
        names += ['backbone.layer3.5.conv2.conv_offset.weight', 'backbone.layer2.1.conv2.conv_offset.bias']
    for k in names:
        a, b = params[k].grad.detach().cpu(), sd[k].grad
        e = rel_l2(a.reshape(b.shape), b)
        print('  fp32 grad %-48s rel L2 %.2e' % (k, e))
        assert e <= (1e-2 if dcn else 2e-3), (k, e)


def test_r101_inference_512_proposals_against_the_oracle():
    """configs[4] at a size the oracle finishes: HTD-R101 simple_test, ONE 256x320 image, nms_post = 512 proposals
    into the RoI head (the oracle's regression branch alone is 0.63 TFLOP on the CPU), hard NMS.  Proposals, both
    stages' logits on the oracle's rois, and the detections."""
    from golden_util import match_detections
    from htd_amd.configs import build_htd_detector, htd_config
    from oracle import detector as D
    dev = torch.device('cuda:0')
    H, W = 256, 320
    imgs, _, _ = demo_inputs(1, H, W, np.random.RandomState(4))
    imgs = (imgs - 0.5) * 4
    metas = [dict(img_shape=(H, W - 5, 3), pad_shape=(H, W, 3), ori_shape=(H, W - 5, 3),
                  scale_factor=np.array([1, 1, 1, 1], dtype=np.float32), flip=False)]
    ocfg = D.htd_config(101)
    ocfg['test_cfg']['rpn'].update(nms_pre=1000, nms_post=512, max_num=512)
    ocfg['test_cfg']['rcnn']['score_thr'] = 0.002
    cfg = htd_config(101, soft_nms=False)
    cfg.test_cfg.rpn.update(nms_pre=1000, nms_post=512, max_num=512)
    cfg.test_cfg.rcnn.score_thr = 0.002
    sd = seeded_state_dict(D.state_shapes(101), prefix='det.')
    with torch.no_grad():
        x = D.extract_feat(sd, T(imgs), ocfg)
        rcls, rreg = D.rpn_forward(sd, x)
        trace = []
        props = D.rpn_get_bboxes(rcls, rreg, metas, ocfg['test_cfg']['rpn'], ocfg, ocfg['strides'], trace=trace)
        tr = {}
        dets = D.roi_head_simple_test(sd, x, props, metas, ocfg, trace=tr)
    assert props[0].shape == (512, 5)
    det = load_seeded_(build_htd_detector(cfg=cfg), 'det.').to(dev).eval()
    rpn = det.rpn_head
    with torch.no_grad():
        feats = det.extract_feat(T(imgs).to(dev))
        # bit-exact index trail on the oracle's RPN logits (as in the R50 fixture test), 512 kept of ~4000 candidates
        rpn.record_trail = True
        try:
            p_inj = rpn.get_bboxes([c.to(dev) for c in rcls], [r.to(dev) for r in rreg], metas)
            order, anchor_ids, n_keep = rpn._last_proposal_trail
        finally:
            rpn.record_trail = False
        assert int(n_keep[0]) == 512
        # (a) candidate selection (per-level sort + top-k on the same logits): identical anchors in identical order
        cand_boxes, cand_scores, cand_ids = rpn._last_candidates
        from oracle import ops as O
        ref_scores = torch.cat([c[0].permute(1, 2, 0).reshape(-1).sigmoid().sort(descending=True, stable=True)[0][:1000] for c in rcls])
        torch.testing.assert_close(cand_scores[0].cpu(), ref_scores, rtol=2e-7, atol=0)          # sigmoid: one ulp
        # (b) decode: the device's expf against the host's, a few ulp of a 320-pixel coordinate
        # (c) NMS + final order on what the device decoded: the oracle's sequential NMS keeps exactly the same rows.
        #     (End to end the two keep lists differ only where an IoU sits within rounding of 0.7: with ~4000 candidates
        #     such a pair exists; the 60-of-300 reference fixture in test_gpu_detector.py is compared end to end.)
        dets_o, keep_o = O.batched_nms(cand_boxes[0].cpu(), cand_scores[0].cpu(), cand_ids.cpu().long(), dict(type='nms', iou_threshold=0.7))
        assert torch.equal(order[0, :512].cpu(), keep_o[:512])
        assert torch.equal(p_inj[0].cpu(), dets_o[:512])
        # (d) end to end the product's keep list and the oracle's differ in a few rows.  Greedy NMS is a function of the
        #     candidates' score ORDER and of the pairwise DECISIONS IoU > 0.7 (rpn_head.py:122-168); the two sides see the same
        #     candidates (same anchors), boxes that differ by ulps (device expf vs host expf in delta2bbox) and scores that
        #     differ by at most one ulp (device sigmoid vs host sigmoid).  Round 2 blamed the boxes; measured here:
        #       * not one pairwise decision differs (were one to differ, its IoU must sit within 1e-5 of the threshold);
        #       * the oracle's NMS on the HOST boxes with the DEVICE scores reproduces the product's kept anchors exactly --
        #         so the one-ulp sigmoid differences, which reorder candidates of (nearly) equal score, are the whole cause.
        _, _, host_boxes, host_scores, host_ids, host_anchors = trace[0]
        # same candidates; positions differ only on a level with fewer anchors than nms_pre, which the reference leaves
        # unsorted (rpn_head.py:124) and the product ranks like the others -- the NMS orders by score either way
        dev_anchors = rpn._last_candidate_anchors[0].cpu()
        perm_d, perm_h = dev_anchors.argsort(), host_anchors.argsort()
        assert torch.equal(dev_anchors[perm_d], host_anchors[perm_h])
        assert torch.equal(cand_ids.cpu().long()[perm_d], host_ids[perm_h])
        dev_scores = cand_scores[0].cpu()[perm_d]
        torch.testing.assert_close(dev_scores, host_scores[perm_h], rtol=2e-7, atol=0)            # one ulp
        dev_boxes, host_boxes, host_ids, anchors = cand_boxes[0].cpu()[perm_d], host_boxes[perm_h], host_ids[perm_h], host_anchors[perm_h]
        np.testing.assert_allclose(dev_boxes.numpy(), host_boxes.numpy(), rtol=1e-5, atol=2e-4)   # (b) a few ulp of 320 px

        def decisions(bx):
            """fp32 `inter / union` of every pair, the arithmetic of the NMS (oracle/c, csrc/nms.hip)"""
            area = (bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1])
            w = (torch.min(bx[:, None, 2], bx[None, :, 2]) - torch.max(bx[:, None, 0], bx[None, :, 0])).clamp(min=0)
            h = (torch.min(bx[:, None, 3], bx[None, :, 3]) - torch.max(bx[:, None, 1], bx[None, :, 1])).clamp(min=0)
            inter = w * h
            return inter / (area[:, None] + area[None, :] - inter)
        for lvl in range(5):
            sel = (host_ids == lvl).nonzero().squeeze(1)
            if sel.numel() < 2:
                continue
            diff = torch.triu((decisions(host_boxes[sel]) > 0.7) != (decisions(dev_boxes[sel]) > 0.7), diagonal=1).nonzero()
            if diff.numel():
                iou64 = decisions(host_boxes[sel].double())[diff[:, 0], diff[:, 1]]
                assert float((iou64 - 0.7).abs().max()) < 1e-5, (lvl, iou64)
        _, keep_mix = O.batched_nms(host_boxes, dev_scores, host_ids, dict(type='nms', iou_threshold=0.7))
        assert torch.equal(anchors[keep_mix[:512]], anchor_ids[0, :512].cpu())                    # the cause, exactly
        mine_ids, ref_ids = anchor_ids[0, :512].cpu().numpy(), trace[0][1].numpy()
        common, ia, ib = np.intersect1d(mine_ids, ref_ids, return_indices=True)
        assert common.size >= 500, common.size
        np.testing.assert_allclose(p_inj[0].cpu().numpy()[ia], props[0].numpy()[ib], rtol=1e-5, atol=2e-4)   # (b)
        gfeat = det.roi_head.glbctx_head(feats)[1]
        for st in (0, 1):
            res = det.roi_head._bbox_forward(st, feats, tr[f'rois{st}'].to(dev), gfeat)
            torch.testing.assert_close(res['cls_score'].cpu(), tr[f'cls{st}'], rtol=0, atol=1e-4)
            torch.testing.assert_close(res['bbox_pred'].cpu(), tr[f'reg{st}'], rtol=0, atol=1e-4)
        res = det.simple_test(T(imgs).to(dev), metas)
    assert len(res) == 1 and len(res[0]) == 80
    mine = np.concatenate([np.concatenate([r, np.full((len(r), 1), c, dtype=np.float32)], 1) for c, r in enumerate(res[0])], 0)
    d, l = dets[0]
    ref = torch.cat([d, l[:, None].float()], 1).numpy()
    assert len(ref) == 100                              # max_per_img reached: the cut is part of what is compared
    match_detections(mine, ref, tol=2e-3)
