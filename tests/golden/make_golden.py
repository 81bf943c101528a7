#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own
Python files from /root/reference (authoring container only; the reference never
travels to the GPU box -- only the .npz data written here does).

How the reference is made importable:
`import mmdet` fails with ModuleNotFoundError('mmcv') -- an ordinary Python error,
not a permission denial -- because the third-party mmcv-full 1.2.1 is absent.  This
script registers a small stand-in for the *third-party* package (Registry,
build_from_cfg, ConvModule = conv->norm->act, init helpers, identity fp16
decorators) and bare namespace packages for `mmdet` whose __path__ points into
/root/reference/mmdet, then imports the reference's real files unchanged.
mmcv.ops.{RoIAlign,nms,batched_nms,soft_nms} are served by oracle/ops.py (the C
restatement); those three operators therefore stay "parity unpinned", everything
else in the fixtures is computed by reference code.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import importlib
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle import ops as oracle_ops  # noqa: E402
sys.path.insert(0, os.path.join(REPO, 'tests'))
from golden_util import demo_inputs, digest, load_seeded_, seeded_tensor  # noqa: E402


# ----------------------------------------------------------------------------
# stand-in for the absent third-party mmcv
# ----------------------------------------------------------------------------
class _Placeholder:
    def __init__(self, *a, **k):
        raise NotImplementedError('mmcv placeholder')


def _permissive(mod):
    def __getattr__(name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name[0].isupper():
            return type(name, (_Placeholder, ), {})

        def _raise(*a, **k):
            raise NotImplementedError(f'mmcv stand-in has no {name}')
        return _raise
    mod.__getattr__ = __getattr__
    return mod


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        self._module_dict[name or cls.__name__] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module
        if isinstance(name, type):  # used as bare decorator
            self._register(name)
            return name

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop('type')
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f'{t} is not in the {registry._name} registry')
    return cls(**args)


class Config(dict):
    """attribute dict that wraps nested dicts and lists recursively"""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = Config.wrap(v)

    @staticmethod
    def wrap(v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v)
        if isinstance(v, (list, tuple)):
            return type(v)(Config.wrap(x) for x in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return Config(dict(self))


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    t = cfg.pop('type')
    rg = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    if t in ('BN', 'BN2d'):
        layer, abbr = nn.BatchNorm2d(num_features, **cfg), 'bn'
    elif t == 'GN':
        layer, abbr = nn.GroupNorm(num_channels=num_features, **cfg), 'gn'
    else:
        raise KeyError(t)
    for p in layer.parameters():
        p.requires_grad = rg
    return abbr + str(postfix), layer


def build_conv_layer(cfg, *args, **kwargs):
    t = 'Conv2d' if cfg is None else dict(cfg).get('type', 'Conv2d')
    if t in ('Conv2d', 'Conv'):
        return nn.Conv2d(*args, **kwargs)
    raise NotImplementedError(f'conv layer {t} needs mmcv.ops')


def kaiming_init(m, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.kaiming_uniform_(m.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    else:
        nn.init.kaiming_normal_(m.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def xavier_init(m, gain=1, bias=0, distribution='normal'):
    if distribution == 'uniform':
        nn.init.xavier_uniform_(m.weight, gain=gain)
    else:
        nn.init.xavier_normal_(m.weight, gain=gain)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def normal_init(m, mean=0, std=1, bias=0):
    nn.init.normal_(m.weight, mean, std)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def constant_init(m, val, bias=0):
    if getattr(m, 'weight', None) is not None:
        nn.init.constant_(m.weight, val)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


class ConvModule(nn.Module):
    """conv -> norm -> act with mmcv's bias='auto' rule and kaiming init."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, with_spectral_norm=False, padding_mode='zeros',
                 order=('conv', 'norm', 'act')):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        kaiming_init(self.conv, a=0, nonlinearity='relu')
        if self.with_norm:
            constant_init(getattr(self, self.norm_name), 1, bias=0)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def _identity_decorator(*dargs, **dkw):
    def deco(fn):
        return fn
    return deco


def install_mmcv_standin():
    mmcv = _permissive(types.ModuleType('mmcv'))
    mmcv.__version__ = '1.2.1'
    mmcv.Config = Config
    mmcv.__path__ = []
    utils = _permissive(types.ModuleType('mmcv.utils'))
    utils.Registry, utils.build_from_cfg = Registry, build_from_cfg
    utils.print_log = lambda *a, **k: None
    utils.get_logger = lambda name, **k: logging.getLogger(name)
    cnn = _permissive(types.ModuleType('mmcv.cnn'))
    cnn.__path__ = []
    for f in (ConvModule, build_conv_layer, build_norm_layer, kaiming_init, xavier_init, normal_init,
              constant_init):
        setattr(cnn, f.__name__, f)
    bricks = _permissive(types.ModuleType('mmcv.cnn.bricks'))
    bricks.ConvModule, bricks.build_conv_layer, bricks.build_norm_layer = \
        ConvModule, build_conv_layer, build_norm_layer
    runner = _permissive(types.ModuleType('mmcv.runner'))
    runner.auto_fp16 = runner.force_fp32 = _identity_decorator
    runner.load_checkpoint = lambda *a, **k: None
    runner.OptimizerHook = type('OptimizerHook', (), {})
    ops = _permissive(types.ModuleType('mmcv.ops'))
    ops.__path__ = []
    ops.RoIAlign = oracle_ops.RoIAlign
    ops.roi_align = oracle_ops.roi_align
    ops.nms, ops.soft_nms, ops.batched_nms = oracle_ops.nms, oracle_ops.soft_nms, oracle_ops.batched_nms
    ops_nms = _permissive(types.ModuleType('mmcv.ops.nms'))
    ops_nms.nms, ops_nms.soft_nms, ops_nms.batched_nms = ops.nms, ops.soft_nms, ops.batched_nms
    parallel = _permissive(types.ModuleType('mmcv.parallel'))
    for name, m in [('mmcv', mmcv), ('mmcv.utils', utils), ('mmcv.cnn', cnn), ('mmcv.cnn.bricks', bricks),
                    ('mmcv.runner', runner), ('mmcv.ops', ops), ('mmcv.ops.nms', ops_nms),
                    ('mmcv.parallel', parallel)]:
        sys.modules[name] = m
    mmcv.utils, mmcv.cnn, mmcv.runner, mmcv.ops, mmcv.parallel = utils, cnn, runner, ops, parallel
    cnn.bricks = bricks
    ops.nms_module = ops_nms


def install_reference_namespace():
    def bare(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, rel)]
        sys.modules[name] = m
        return m
    mmdet = bare('mmdet', 'mmdet')
    core = bare('mmdet.core', 'mmdet/core')
    bare('mmdet.core.post_processing', 'mmdet/core/post_processing')
    mutils = bare('mmdet.utils', 'mmdet/utils')
    mutils.get_root_logger = lambda *a, **k: logging.getLogger('mmdet')
    bare('mmdet.models', 'mmdet/models')
    for sub in ('backbones', 'necks', 'dense_heads', 'detectors', 'roi_heads'):
        bare(f'mmdet.models.{sub}', f'mmdet/models/{sub}')
    bare('mmdet.models.roi_heads.bbox_heads', 'mmdet/models/roi_heads/bbox_heads')
    bare('mmdet.models.roi_heads.roi_extractors', 'mmdet/models/roi_heads/roi_extractors')
    mmdet.core = core
    for name in ('mmdet.core.anchor', 'mmdet.core.bbox', 'mmdet.core.utils.misc',
                 'mmdet.core.post_processing.bbox_nms', 'mmdet.core.post_processing.merge_augs'):
        try:
            m = importlib.import_module(name)
        except Exception as e:  # utils/__init__ pulls dist_utils -> mmcv.runner (fine)
            print('note:', name, e)
            continue
        for k, v in vars(m).items():
            if not k.startswith('_'):
                setattr(core, k, v)
    bbox = importlib.import_module('mmdet.core.bbox')
    for k in getattr(bbox, '__all__', []):
        setattr(core, k, getattr(bbox, k))


def ref(name):
    return importlib.import_module(name)


def load_cfg(path):
    ns = {}
    exec(open(os.path.join(REF, path)).read(), ns)
    return ns


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)')


# ----------------------------------------------------------------------------
# fixtures
# ----------------------------------------------------------------------------
def gen_box_math():
    """anchors, IoU, assigner, coder, sampler: reference core/ functions on seeded inputs,
    including the reference tests' own known-answer cases."""
    core = sys.modules['mmdet.core']
    AnchorGenerator = core.AnchorGenerator
    ag = AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    sizes = [(8, 12), (4, 6), (2, 3), (1, 2), (1, 1)]
    anchors = ag.grid_anchors(sizes, device='cpu')
    flags = ag.valid_flags(sizes, (30, 45, 3), device='cpu')
    out = {f'anchors{i}': a for i, a in enumerate(anchors)}
    out.update({f'flags{i}': f for i, f in enumerate(flags)})
    out.update({f'base{i}': b for i, b in enumerate(ag.base_anchors)})
    out['sizes'] = np.array(sizes)
    out['pad_shape'] = np.array([30, 45, 3])
    # reference KAT tests/test_anchor.py:22-40
    kat = AnchorGenerator([10], [1.], [1.], [10]).grid_anchors([(2, 2)], device='cpu')[0]
    out['kat_anchor'] = kat
    kat2 = AnchorGenerator([(10, 20)], [1.], [1.], [10]).grid_anchors([(2, 2)], device='cpu')[0]
    out['kat_anchor_xy'] = kat2
    npz('anchors', **out)

    g = torch.Generator().manual_seed(11)
    def boxes(n, s=100.):
        xy = torch.rand(n, 2, generator=g) * s
        wh = torch.rand(n, 2, generator=g) * s * 0.5 + 1
        return torch.cat([xy, xy + wh], 1)
    b1, b2 = boxes(37), boxes(9)
    coder = ref('mmdet.core.bbox.coder.delta_xywh_bbox_coder')
    core.bbox2delta, core.delta2bbox = coder.bbox2delta, coder.delta2bbox
    iou = core.bbox_overlaps(b1, b2)
    iof = core.bbox_overlaps(b1, b2, mode='iof')
    iou_al = core.bbox_overlaps(b1[:9], b2, is_aligned=True)
    deltas = core.bbox2delta(b1[:9], b2, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2))
    rnd = torch.randn(37, 4, generator=g)
    dec = core.delta2bbox(b1, rnd, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2), max_shape=(80, 90, 3))
    dec_noclip = core.delta2bbox(b1, rnd * 10, (0., 0., 0., 0.), (1., 1., 1., 1.))
    # reference docstring KAT core/bbox/coder/delta_xywh_bbox_coder.py:156-169
    kat_rois = torch.Tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    kat_deltas = torch.Tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.],
                               [0.7, -1.9, -0.5, 0.3]])
    kat_dec = core.delta2bbox(kat_rois, kat_deltas, max_shape=(32, 32))
    npz('box_math', b1=b1, b2=b2, iou=iou, iof=iof, iou_aligned=iou_al, deltas=deltas, rnd=rnd,
        dec=dec, dec_noclip=dec_noclip, kat_rois=kat_rois, kat_deltas=kat_deltas, kat_dec=kat_dec)

    # assigner: reference KATs (tests/test_assigner.py:14-35 and ignore/empty cases) + random
    MaxIoUAssigner = core.MaxIoUAssigner
    out = {}
    bb = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42]])
    gg = torch.FloatTensor([[0, 0, 10, 9], [0, 10, 10, 19]])
    r = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5).assign(bb, gg, gt_labels=torch.LongTensor([2, 3]))
    out.update(kat_bboxes=bb, kat_gts=gg, kat_gt_inds=r.gt_inds, kat_labels=r.labels,
               kat_max_overlaps=r.max_overlaps)
    for tag, kw in [('rpn', dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True,
                                 ignore_iof_thr=-1)),
                    ('rcnn', dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False,
                                  ignore_iof_thr=-1))]:
        pb, gb = boxes(300), boxes(6)
        gl = torch.randint(0, 80, (6, ), generator=g)
        r = MaxIoUAssigner(**kw).assign(pb, gb, gt_labels=gl)
        out.update({f'{tag}_bboxes': pb, f'{tag}_gts': gb, f'{tag}_gt_labels': gl, f'{tag}_gt_inds': r.gt_inds,
                    f'{tag}_max_overlaps': r.max_overlaps, f'{tag}_labels': r.labels})
    r = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5).assign(bb, torch.empty(0, 4))
    out['empty_gt_inds'] = r.gt_inds
    npz('assigner', **out)

    # sampler with a replayed CPU generator (random_sampler.py:54 torch.randperm)
    RandomSampler = core.RandomSampler
    pb, gb = boxes(400), boxes(5)
    gl = torch.randint(0, 80, (5, ), generator=g)
    assign = MaxIoUAssigner(pos_iou_thr=0.3, neg_iou_thr=0.3, min_pos_iou=0.3,
                            match_low_quality=False).assign(pb, gb, gt_labels=gl)
    torch.manual_seed(123)
    s = RandomSampler(num=64, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True) \
        .sample(assign, pb, gb, gl)
    npz('sampler', bboxes=pb, gts=gb, gt_labels=gl, seed=123, pos_inds=s.pos_inds, neg_inds=s.neg_inds,
        pos_is_gt=s.pos_is_gt, pos_assigned_gt_inds=s.pos_assigned_gt_inds, pos_gt_labels=s.pos_gt_labels,
        pos_bboxes=s.pos_bboxes, neg_bboxes=s.neg_bboxes)


def put_digest(out, key, t):
    sums, sample = digest(t)
    out[key + '.sums'] = sums
    out[key + '.sample'] = sample


def gen_heads():
    """HTD-specific numerics that no reference test pins: SFA, PGraph (HTDBBoxHead fwd+bwd),
    BA (AdptRoIExtractor), stage-1 head, losses.  Weights and big inputs are seeded
    (tests/golden_util.py), outputs are stored whole, gradients as digests."""
    ref('mmdet.models.losses')
    ref('mmdet.models.roi_heads.bbox_heads.bbox_head')
    ref('mmdet.models.roi_heads.bbox_heads.convfc_bbox_head')
    gch = ref('mmdet.models.roi_heads.bbox_heads.global_context_head')
    ref('mmdet.models.roi_heads.bbox_heads.htd_bbox_head')
    ref('mmdet.models.roi_heads.roi_extractors.single_level_roi_extractor')
    adp = ref('mmdet.models.roi_heads.roi_extractors.adaptative_roi_extractor')
    builder = ref('mmdet.models.builder')
    cfg = load_cfg('configs/htd/htd_resnet50_1x.py')
    gen = torch.Generator().manual_seed(5)

    # ---- SFA (global_context_head.py:382-401) ----
    sfa = gch.GlobalContextHead(num_ins=5, num_convs=4, in_channels=256, conv_out_channels=256,
                                num_classes=81, loss_weight=3.0)
    load_seeded_(sfa, 'sfa.')
    p6 = seeded_tensor('sfa.p6', (2, 256, 5, 7)).requires_grad_()
    mc_pred, gfeat = sfa([p6])
    labels = [torch.tensor([3, 3, 17]), torch.tensor([80 - 1, 0])]
    loss = sfa.loss(mc_pred, labels)
    loss.backward()
    out = dict(mc_pred=mc_pred, global_feat=gfeat, loss=loss, labels0=labels[0], labels1=labels[1])
    put_digest(out, 'grad_p6', p6.grad)
    put_digest(out, 'grad_fc_w', sfa.fc.weight.grad)
    put_digest(out, 'grad_conv0_w', sfa.convs[0].conv.weight.grad)
    npz('sfa', **out)

    # ---- stage-1 head + loss (convfc_bbox_head.py:135-173, bbox_head.py:142-186) ----
    head0 = builder.build_head(dict(cfg['model']['roi_head']['bbox_head'][0]))
    load_seeded_(head0, 'head0.')
    x = seeded_tensor('head0.x', (24, 256, 7, 7)).requires_grad_()
    cls, reg = head0(x)
    lab = torch.randint(0, 81, (24, ), generator=gen)
    lab[:6] = torch.randint(0, 80, (6, ), generator=gen)
    lw = torch.ones(24)
    bt = torch.randn(24, 4, generator=gen)
    bw = (lab < 80).float()[:, None].expand(24, 4).contiguous()
    rois = torch.cat([torch.zeros(24, 1), torch.rand(24, 4, generator=gen) * 50], 1)
    losses = head0.loss(cls, reg, rois, lab, lw, bt, bw)
    (losses['loss_cls'] + losses['loss_bbox']).backward()
    out = dict(cls=cls, reg=reg, labels=lab, label_weights=lw, bbox_targets=bt, bbox_weights=bw,
               loss_cls=losses['loss_cls'], loss_bbox=losses['loss_bbox'], acc=losses['acc'])
    put_digest(out, 'grad_x', x.grad)
    put_digest(out, 'grad_fc_cls_w', head0.fc_cls.weight.grad)
    npz('stage1_head', **out)

    # ---- PGraph: HTDBBoxHead.forward + backward (htd_bbox_head.py:157-230) ----
    h1 = builder.build_head(dict(cfg['model']['roi_head']['bbox_head'][1]))
    load_seeded_(h1, 'head1.')
    N, Np = 40, 10

    def mk(n, img):  # rois spanning all four pyramid levels, some overlapping, some disjoint
        s = torch.tensor([20., 60., 130., 260., 500.])[torch.randint(0, 5, (n, ), generator=gen)]
        cx = torch.rand(n, generator=gen) * 600 + 100
        cy = torch.rand(n, generator=gen) * 400 + 100
        a = torch.exp((torch.rand(n, generator=gen) - 0.5))
        w, h = s * a, s / a
        return torch.stack([torch.full((n, ), float(img)), cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    rois = torch.cat([mk(22, 0), mk(18, 1)])
    pos_idx = torch.cat([torch.arange(0, 6), torch.arange(22, 26)])
    pos_rois = rois[pos_idx]
    x_cls = seeded_tensor('head1.x_cls', (N, 256, 7, 7)).requires_grad_()
    enhanced = seeded_tensor('head1.enhanced', (Np, 256, 7, 7)).requires_grad_()
    gfeat = seeded_tensor('head1.gfeat', (2, 256, 1, 1)).requires_grad_()
    x_reg = x_cls[pos_idx]
    cls, reg = h1(x_cls, x_reg, [None] * 4, rois, head0.fc_cls, enhanced, pos_rois, gfeat)
    gcls = seeded_tensor('head1.gcls', cls.shape)
    greg = seeded_tensor('head1.greg', reg.shape)
    head0.zero_grad()
    ((cls * gcls).sum() + (reg * greg).sum()).backward()
    out = dict(rois=rois, pos_idx=pos_idx, cls=cls, reg=reg, target_lvls=h1.map_roi_levels(rois, 4))
    for key, t in [('grad_x_cls', x_cls.grad), ('grad_enhanced', enhanced.grad), ('grad_global', gfeat.grad),
                   ('grad_fc_cls_0_w', head0.fc_cls.weight.grad), ('grad_fc_cls_0_b', head0.fc_cls.bias.grad),
                   ('grad_graph0_w', h1.graph_lvl0_cls.weight.grad), ('grad_graph3_w', h1.graph_lvl3_cls.weight.grad),
                   ('grad_fcs0_w', h1.fcs[0].weight.grad), ('grad_conv0_w', h1.convs[0].conv.weight.grad),
                   ('grad_gn0_w', h1.convs[0].gn.weight.grad), ('grad_fc_reg_w', h1.fc_reg.weight.grad)]:
        put_digest(out, key, t)
    npz('pgraph', **out)

    # ---- BA: AdptRoIExtractor (adaptative_roi_extractor.py:49-91; RoIAlign = oracle/ops.py) ----
    ecfg = dict(cfg['model']['roi_head']['bbox_roi_extractor'][1])
    ecfg.pop('type')
    ba = adp.AdptRoIExtractor(**ecfg)
    load_seeded_(ba, 'ba.')
    feats = [seeded_tensor(f'ba.feat{i}', (2, 256, 64 // s, 96 // s)).requires_grad_()
             for i, s in enumerate((1, 2, 4, 8))]
    brois = torch.tensor([[0, 10., 12., 90., 70.], [0, 100., 30., 180., 200.], [1, 5., 5., 40., 33.],
                          [1, 0., 0., 383., 255.], [1, 200., 100., 260., 130.]])
    o = ba(feats, brois)
    go = seeded_tensor('ba.go', o.shape)
    (o * go).sum().backward()
    out = dict(rois=brois, out=o)
    for i, f in enumerate(feats):
        put_digest(out, f'grad_feat{i}', f.grad)
    put_digest(out, 'grad_conv1_w', ba.conv1.weight.grad)
    put_digest(out, 'grad_conv2_w', ba.conv2.weight.grad)
    npz('ba', **out)


def small_model_cfg(cfg):
    """HTD-R50 config with nothing changed but the proposal/sample counts, so the whole
    reference forward_train runs in seconds on a 128x160 image pair."""
    model = cfg['model']
    model['pretrained'] = None
    train_cfg = Config(cfg['train_cfg'])
    test_cfg = Config(cfg['test_cfg'])
    train_cfg.rpn_proposal.nms_pre = 200
    train_cfg.rpn_proposal.nms_post = 100
    train_cfg.rpn_proposal.max_num = 100
    for r in train_cfg.rcnn:
        r.sampler.num = 48
    test_cfg.rpn.nms_pre = 100
    test_cfg.rpn.nms_post = 60
    test_cfg.rpn.max_num = 60
    test_cfg.rcnn.score_thr = 0.001
    return model, train_cfg, test_cfg


def gen_detector():
    """Whole-path fixture: the reference FasterRCNN+HTDRoIHead forward_train (losses and a
    few parameter gradients) and simple_test on 2 synthetic images, sampler RNG replayed
    from torch.manual_seed.  Weights are NOT stored (74 M params): they are re-created by
    tests/golden_util.py::load_seeded_(det, 'det.')."""
    for m in ('mmdet.models.losses', 'mmdet.models.backbones.resnet', 'mmdet.models.necks.fpn',
              'mmdet.models.dense_heads.anchor_head', 'mmdet.models.dense_heads.rpn_head',
              'mmdet.models.roi_heads.base_roi_head', 'mmdet.models.roi_heads.bbox_heads.bbox_head',
              'mmdet.models.roi_heads.bbox_heads.convfc_bbox_head',
              'mmdet.models.roi_heads.bbox_heads.global_context_head',
              'mmdet.models.roi_heads.bbox_heads.htd_bbox_head',
              'mmdet.models.roi_heads.roi_extractors.single_level_roi_extractor',
              'mmdet.models.roi_heads.roi_extractors.adaptative_roi_extractor',
              'mmdet.models.roi_heads.htd_roi_head', 'mmdet.models.detectors.base',
              'mmdet.models.detectors.two_stage', 'mmdet.models.detectors.faster_rcnn'):
        ref(m)
    builder = ref('mmdet.models.builder')
    cfg = load_cfg('configs/htd/htd_resnet50_1x.py')
    model_cfg, train_cfg, test_cfg = small_model_cfg(cfg)
    torch.manual_seed(0)
    det = builder.build_detector(model_cfg, train_cfg=train_cfg, test_cfg=test_cfg)
    det.init_weights(None)
    load_seeded_(det, 'det.')
    det.train()
    rng = np.random.RandomState(0)
    B, H, W = 2, 128, 160
    imgs, gts, labels = demo_inputs(B, H, W, rng)
    imgs = (imgs - 0.5) * 4
    metas = [dict(img_shape=(H, W - 3, 3), pad_shape=(H, W, 3), ori_shape=(H, W - 3, 3),
                  scale_factor=np.array([1, 1, 1, 1], dtype=np.float32), flip=False) for _ in range(B)]
    gts = [np.minimum(g, np.array([W - 3, H, W - 3, H], dtype=np.float32)) for g in gts]
    img_t = torch.from_numpy(imgs)
    # trail of the RoI head: what each stage was fed and what it answered (north_star: logits within 1e-4)
    trail = {}
    orig_bf = det.roi_head._bbox_forward

    def rec_bf(stage, x, rois, *a, **k):
        r = orig_bf(stage, x, rois, *a, **k)
        trail[stage] = (rois.detach().clone(), r['cls_score'].detach().clone(), r['bbox_pred'].detach().clone())
        return r
    det.roi_head._bbox_forward = rec_bf
    torch.manual_seed(77)
    losses = det.forward_train(img_t, metas, [torch.from_numpy(g) for g in gts],
                               [torch.from_numpy(l) for l in labels])
    loss, log_vars = det._parse_losses(losses)
    det.zero_grad()
    loss.backward()
    sd = dict(det.named_parameters())
    grads = {}
    for k in ('backbone.layer2.0.conv1.weight', 'backbone.layer4.2.bn3.weight', 'backbone.layer3.1.bn2.bias',
              'neck.lateral_convs.0.conv.weight', 'neck.fpn_convs.3.conv.bias', 'rpn_head.rpn_conv.weight',
              'rpn_head.rpn_reg.bias', 'roi_head.bbox_head.0.fc_cls.weight',
              'roi_head.bbox_head.0.shared_fcs.0.bias', 'roi_head.bbox_head.1.fc_reg.weight',
              'roi_head.bbox_head.1.graph_lvl0_cls.weight', 'roi_head.bbox_head.1.convs.1.gn.weight',
              'roi_head.bbox_roi_extractor.1.conv1.weight', 'roi_head.glbctx_head.fc.weight'):
        gr = sd[k].grad
        put_digest(grads, 'grad.' + k, gr if gr is not None else torch.zeros_like(sd[k]))
    out = dict(H=H, W=W, seed_sampler=77, img_w=W - 3)
    for st in (0, 1):
        out[f'train_s{st}_rois'], out[f'train_s{st}_cls'], out[f'train_s{st}_reg'] = trail[st]
    for i in range(B):
        out[f'gt{i}'] = gts[i]
        out[f'label{i}'] = labels[i]
    for k, v in log_vars.items():
        out['loss.' + k] = np.float64(v)
    out.update(grads)
    # inference
    det.eval()
    # trail of the proposal stage: the NMS call of every image (rpn_head.py:166-167) with its candidates and keep
    rpn_mod = sys.modules['mmdet.models.dense_heads.rpn_head']
    nms_calls = []
    orig_nms = rpn_mod.batched_nms

    def rec_nms(boxes, scores, ids, cfg):
        dets, keep = orig_nms(boxes, scores, ids, cfg)
        nms_calls.append((boxes.clone(), scores.clone(), ids.clone(), keep.clone()))
        return dets, keep
    rpn_mod.batched_nms = rec_nms
    with torch.no_grad():
        feats = det.extract_feat(img_t)
        rpn_cls, rpn_reg = det.rpn_head(feats)
        props = det.rpn_head.get_bboxes(rpn_cls, rpn_reg, metas)      # = simple_test_rpn (rpn_test_mixin.py:24-37)
        res = det.roi_head.simple_test(feats, props, metas, rescale=False)
    rpn_mod.batched_nms = orig_nms
    assert len(nms_calls) == B
    for l in range(len(rpn_cls)):
        out[f'rpn_cls{l}'] = rpn_cls[l]
        out[f'rpn_reg{l}'] = rpn_reg[l]
    level_off = np.cumsum([0] + [int(c.shape[1] * c.shape[2] * c.shape[3]) for c in rpn_cls])
    for i in range(B):
        boxes_i, scores_i, ids_i, keep_i = nms_calls[i]
        # anchor identity of every candidate: the same sort the reference ran (rpn_head.py:135-137), per level
        flat_ids = []
        for l in range(len(rpn_cls)):
            sc = rpn_cls[l][i].permute(1, 2, 0).reshape(-1).sigmoid()
            if test_cfg.rpn.nms_pre > 0 and sc.shape[0] > test_cfg.rpn.nms_pre:
                ranked, rank_inds = sc.sort(descending=True)
                rank_inds, ranked = rank_inds[:test_cfg.rpn.nms_pre], ranked[:test_cfg.rpn.nms_pre]
            else:
                rank_inds, ranked = torch.arange(sc.shape[0]), sc
            assert torch.equal(ranked, scores_i[ids_i == l])
            flat_ids.append(rank_inds + int(level_off[l]))
        flat_ids = torch.cat(flat_ids)
        keep_i = keep_i[:test_cfg.rpn.nms_post]
        out[f'test_keep{i}'] = keep_i                         # rows of the level-concatenated candidate list, kept order
        out[f'test_prop_anchor{i}'] = flat_ids[keep_i]        # ... and which anchor of the image each of them is
        assert torch.equal(boxes_i[keep_i], props[i][:, :4])
    for st in (0, 1):
        out[f'test_s{st}_rois'], out[f'test_s{st}_cls'], out[f'test_s{st}_reg'] = trail[st]
    for i in range(B):
        out[f'test_props{i}'] = props[i]
        out[f'test_dets{i}'] = np.concatenate([np.concatenate([r, np.full((len(r), 1), c, dtype=np.float32)], 1)
                                               for c, r in enumerate(res[i])], 0)
    for i, f in enumerate(feats):
        out[f'feat{i}_sum'] = f.double().sum()
        out[f'feat{i}_abs'] = f.double().abs().sum()
    npz('detector', **out)


def gen_resnext():
    """A ResNeXt stage of the reference (backbones/resnext.py Bottleneck + utils/res_layer.py ResLayer; groups = 8,
    base_width = 4, two blocks, stride 2) on seeded weights: output and gradient digests."""
    ref('mmdet.models.backbones.resnet')
    rx = ref('mmdet.models.backbones.resnext')
    ResLayer = ref('mmdet.models.utils.res_layer').ResLayer
    torch.manual_seed(0)
    layer = ResLayer(block=rx.Bottleneck, inplanes=64, planes=64, num_blocks=2, stride=2, groups=8, base_width=4,
                     base_channels=64, norm_cfg=dict(type='BN', requires_grad=True))
    load_seeded_(layer, 'xblk.')
    layer.eval()
    x = torch.randn(2, 64, 12, 14, generator=torch.Generator().manual_seed(1)).requires_grad_()
    y = layer(x)
    go = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(go)
    out = dict(y=y.detach().numpy())
    put_digest(out, 'gx', x.grad)
    for k, p_ in layer.named_parameters():
        put_digest(out, 'grad.' + k, p_.grad)
    out['conv2_shape'] = np.array(layer[0].conv2.weight.shape)
    npz('resnext_stage', **out)
    print('resnext_stage: y', tuple(y.shape), 'conv2', tuple(layer[0].conv2.weight.shape))


def pipeline_samples():
    """Decoded-image stand-ins + annotations shared by the generator and the tests (seeded)."""
    out = []
    for s, (h, w) in enumerate([(120, 160), (200, 150), (97, 333), (64, 64)]):
        rs = np.random.RandomState(s)
        xy = rs.uniform(0, [w - 8, h - 8], (5, 2))
        boxes = np.concatenate([xy, xy + rs.uniform(4, 60, (5, 2))], 1).astype(np.float32)
        img = np.random.RandomState(100 + s).randint(0, 256, (h, w, 3)).astype(np.uint8)
        out.append((img, boxes, rs.randint(0, 80, 5).astype(np.int64)))
    return out


def gen_pipeline():
    """The reference's own transform classes (datasets/pipelines/transforms.py Resize / RandomFlip / Normalize / Pad,
    formating.py, test_time_aug.py, compose.py) on seeded samples.  The mmcv image functions they call (imrescale,
    imflip, imnormalize, impad_to_multiple: cv2-backed, cv2 is not installed) are served by oracle/pipeline.py, so
    this fixture pins the box / meta / random-number logic to the reference and the pixels to that restatement."""
    from oracle import pipeline as OP
    mmcv = sys.modules['mmcv']
    mmcv.is_list_of = lambda seq, t: isinstance(seq, list) and all(isinstance(x, t) for x in seq)
    mmcv.is_str = lambda x: isinstance(x, str)

    def imrescale(img, scale, return_scale=False, interpolation='bilinear', backend=None):
        out, f = OP.imrescale(img, scale)
        return (out, f) if return_scale else out

    def imresize(img, size, return_scale=False, interpolation='bilinear', out=None, backend=None):
        h, w = img.shape[:2]
        r = OP.imresize_bilinear_u8(img, size)
        return (r, size[0] / w, size[1] / h) if return_scale else r
    mmcv.imrescale, mmcv.imresize = imrescale, imresize
    mmcv.imflip = lambda img, direction='horizontal': OP.imflip(img, direction)
    mmcv.imnormalize = lambda img, mean, std, to_rgb=True: OP.imnormalize(img, mean, std, to_rgb)
    mmcv.impad_to_multiple = lambda img, divisor, pad_val=0: OP.impad_to_multiple(img, divisor, pad_val)
    core = sys.modules['mmdet.core']
    core.PolygonMasks = type('PolygonMasks', (), {})
    ev = types.ModuleType('mmdet.core.evaluation')
    ev.__path__ = [os.path.join(REF, 'mmdet/core/evaluation')]
    sys.modules['mmdet.core.evaluation'] = ev
    ds = types.ModuleType('mmdet.datasets')
    ds.__path__ = [os.path.join(REF, 'mmdet/datasets')]
    sys.modules['mmdet.datasets'] = ds
    pl = types.ModuleType('mmdet.datasets.pipelines')
    pl.__path__ = [os.path.join(REF, 'mmdet/datasets/pipelines')]
    sys.modules['mmdet.datasets.pipelines'] = pl
    tr = ref('mmdet.datasets.pipelines.transforms')
    tta = ref('mmdet.datasets.pipelines.test_time_aug')
    comp = ref('mmdet.datasets.pipelines.compose')
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    pipe = comp.Compose([tr.Resize(img_scale=(1333, 800), keep_ratio=True), tr.RandomFlip(flip_ratio=0.5),
                         tr.Normalize(**norm), tr.Pad(size_divisor=32)])
    ms = comp.Compose([tr.Resize(img_scale=[(1600, 400), (1600, 1400)], multiscale_mode='range', keep_ratio=True),
                       tr.RandomFlip(flip_ratio=0.5)])
    out = {}
    np.random.seed(7)
    for i, (img, boxes, labels) in enumerate(pipeline_samples()):
        r = pipe(dict(img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=['img'], gt_bboxes=boxes.copy(),
                      bbox_fields=['gt_bboxes'], mask_fields=[], seg_fields=[]))
        out[f's{i}.img_shape'] = np.array(r['img_shape'])
        out[f's{i}.pad_shape'] = np.array(r['pad_shape'])
        out[f's{i}.scale_factor'] = r['scale_factor']
        out[f's{i}.flip'] = np.array(int(bool(r['flip'])))
        out[f's{i}.gt_bboxes'] = r['gt_bboxes']
        put_digest(out, f's{i}.img', torch.from_numpy(np.ascontiguousarray(r['img'])))
    np.random.seed(3)
    for i, (img, boxes, labels) in enumerate(pipeline_samples()[:2]):
        r = ms(dict(img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=['img'], gt_bboxes=boxes.copy(),
                    bbox_fields=['gt_bboxes'], mask_fields=[], seg_fields=[]))
        out[f'm{i}.img_shape'] = np.array(r['img_shape'])
        out[f'm{i}.flip'] = np.array(int(bool(r['flip'])))
        out[f'm{i}.gt_bboxes'] = r['gt_bboxes']
    t = tta.MultiScaleFlipAug(transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                                          dict(type='Pad', size_divisor=32)],
                              img_scale=[(1333, 800), (1000, 600)], flip=True)
    img = pipeline_samples()[0][0]
    r = t(dict(img=img, img_shape=img.shape, ori_shape=img.shape, img_fields=['img'], bbox_fields=[], mask_fields=[],
               seg_fields=[]))
    out['tta.img_shapes'] = np.array([a for a in r['img_shape']])
    out['tta.pad_shapes'] = np.array([a for a in r['pad_shape']])
    out['tta.flips'] = np.array([int(bool(f)) for f in r['flip']])
    npz('pipeline', **out)
    print('pipeline: flips', [int(out[f's{i}.flip']) for i in range(4)], 'tta', out['tta.img_shapes'].tolist())


def aug_inputs():
    """One image under two test-time augmentations (scale 1.0 unflipped, scale 1.25 flipped): the recipe shared by
    this generator, tests/test_oracle_golden.py and tests/test_gpu_detector.py."""
    rs = np.random.RandomState(5)
    imgs, metas = [], []
    for (H, W), (h, w), sf, flip in [((160, 224), (150, 210), 1.0, False), ((192, 256), (188, 256), 1.25, True)]:
        im = ((rs.rand(1, 3, H, W) - 0.5) * 4).astype(np.float32)
        im[:, :, h:] = 0
        im[:, :, :, w:] = 0
        imgs.append(im)
        metas.append([dict(img_shape=(h, w, 3), pad_shape=(H, W, 3), ori_shape=(150, 210, 3),
                           scale_factor=np.array([sf] * 4, dtype=np.float32), flip=flip,
                           flip_direction='horizontal' if flip else None)])
    return imgs, metas


def gen_aug_test():
    """The reference's own test-time-augmentation path -- TwoStageDetector.aug_test -> RPNTestMixin.aug_test_rpn ->
    merge_aug_proposals, HTDRoIHead.aug_test -> merge_aug_bboxes -> multiclass_nms -- on the seeded detector."""
    for m in ('mmdet.models.losses', 'mmdet.models.backbones.resnet', 'mmdet.models.necks.fpn',
              'mmdet.models.dense_heads.anchor_head', 'mmdet.models.dense_heads.rpn_test_mixin',
              'mmdet.models.dense_heads.rpn_head',
              'mmdet.models.roi_heads.base_roi_head', 'mmdet.models.roi_heads.bbox_heads.bbox_head',
              'mmdet.models.roi_heads.bbox_heads.convfc_bbox_head',
              'mmdet.models.roi_heads.bbox_heads.global_context_head',
              'mmdet.models.roi_heads.bbox_heads.htd_bbox_head',
              'mmdet.models.roi_heads.roi_extractors.single_level_roi_extractor',
              'mmdet.models.roi_heads.roi_extractors.adaptative_roi_extractor',
              'mmdet.models.roi_heads.htd_roi_head', 'mmdet.models.detectors.base',
              'mmdet.models.detectors.two_stage', 'mmdet.models.detectors.faster_rcnn'):
        ref(m)
    builder = ref('mmdet.models.builder')
    cfg = load_cfg('configs/htd/htd_resnet50_1x.py')
    model_cfg, train_cfg, test_cfg = small_model_cfg(cfg)
    torch.manual_seed(0)
    det = builder.build_detector(model_cfg, train_cfg=train_cfg, test_cfg=test_cfg)
    det.init_weights(None)
    load_seeded_(det, 'det.')
    det.eval()
    imgs, metas = aug_inputs()
    imgs = [torch.from_numpy(i) for i in imgs]
    with torch.no_grad():
        feats = det.extract_feats(imgs)
        props = det.rpn_head.aug_test_rpn(feats, metas)
        res = det.aug_test(imgs, metas)
    assert len(props) == 1 and len(res) == 1
    dets = np.concatenate([np.concatenate([r, np.full((len(r), 1), c, dtype=np.float32)], 1)
                           for c, r in enumerate(res[0])], 0)
    npz('aug_test', proposals=props[0].numpy(), dets=dets)
    print('aug_test: proposals', tuple(props[0].shape), 'detections', dets.shape)


def main():
    torch.set_num_threads(8)
    install_mmcv_standin()
    install_reference_namespace()
    which = sys.argv[1:] or ['box', 'heads', 'detector']
    if 'box' in which:
        gen_box_math()
    if 'heads' in which:
        gen_heads()
    if 'detector' in which:
        gen_detector()
    if 'aug' in which:
        gen_aug_test()
    if 'resnext' in which:
        gen_resnext()
    if 'pipeline' in which:
        gen_pipeline()


if __name__ == '__main__':
    main()
