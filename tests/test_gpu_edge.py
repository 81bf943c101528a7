"""Configurations and corner cases of the path that the parity tests do not reach: the R101-DCN config, soft-NMS
inference, an inference batch without detections, a single tiny image.  Each must run and stay finite."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small(cfg):
    cfg.train_cfg.rpn_proposal.update(nms_pre=300, nms_post=200, max_num=200)
    for r in cfg.train_cfg.rcnn:
        r.sampler.num = 64
    cfg.test_cfg.rpn.update(nms_pre=200, nms_post=100, max_num=100)
    return cfg


@pytest.mark.parametrize('bf16', [False, True])
def test_r101_dcn_train_step(bf16):
    """BASELINE configs[3] architecture (R101-DCN), fp32 and with the bf16 precision map."""
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.runner import Trainer, synthetic_batch
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = build_htd_detector(cfg=_small(htd_config(101, dcn=True)), bf16=bf16).to(dev).train()
    tr = Trainer(model, lr=0.01)
    data = synthetic_batch(2, 256, 320, 311, device=dev, seed=1)
    for _ in range(2):
        out = tr.train_step(data)
    assert torch.isfinite(out['loss'].detach()).item()
    assert torch.isfinite(tr.flat.flat).all().item()


@pytest.mark.parametrize('soft_nms,score_thr', [(True, 0.05), (False, 0.999999)])
def test_inference_variants(soft_nms, score_thr):
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.runner import synthetic_batch
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = _small(htd_config(50, soft_nms=soft_nms))
    cfg.test_cfg.rcnn.score_thr = score_thr
    model = build_htd_detector(cfg=cfg).to(dev).eval()
    data = synthetic_batch(2, 256, 320, 311, device=dev, seed=1)
    with torch.no_grad():
        res = model.simple_test(data['img'], data['img_metas'])
    assert len(res) == 2 and all(len(r) == 80 for r in res)
    assert all(c.shape[1] == 5 for r in res for c in r)


def test_single_tiny_image_train_step():
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.runner import Trainer, synthetic_batch
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = build_htd_detector(cfg=htd_config(50)).to(dev).train()
    tr = Trainer(model, lr=0.01)
    out = tr.train_step(synthetic_batch(1, 128, 160, 150, device=dev, seed=2))
    assert torch.isfinite(out['loss'].detach()).item()


def test_training_reduces_the_loss_on_a_fixed_batch():
    """End-to-end sanity of everything between the losses and the parameters (static-shape path, fused backward,
    gradient sinks, chained pyramid gradients, fused SGD with warm-up): 30 steps on one fixed batch must bring the
    loss down clearly and keep every parameter finite."""
    from htd_amd.configs import build_htd_detector, htd_config
    from htd_amd.runner import Trainer, WarmupStepLR, synthetic_batch
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = build_htd_detector(cfg=_small(htd_config(50))).to(dev).train()
    tr = Trainer(model, schedule=WarmupStepLR(0.005, warmup_iters=10, warmup_ratio=0.1))
    data = synthetic_batch(2, 256, 320, 311, device=dev, seed=5)
    losses = []
    for _ in range(30):
        losses.append(float(tr.train_step(data)['loss'].detach()))
    first, last = sum(losses[:3]) / 3, sum(losses[-3:]) / 3
    assert all(l == l and l < 1e4 for l in losses), losses
    assert last < 0.7 * first, (first, last, losses)
    assert torch.isfinite(tr.flat.flat).all().item()
