"""optim.HdAdamW (csrc/optim.hip, hd_adamw_multi): one launch per parameter group, bf16 shadows written in the same pass,
against torch.optim.AdamW step for step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32, 3, 3), (128,), (33, 7), (1,), (256, 64, 1, 1), (5, 3, 6, 6), (70001,)]
    ps = []
    for i, s in enumerate(shapes):
        t = torch.randn(s, generator=g).to(dev)
        if len(s) == 4 and i % 2 == 0:
            t = t.contiguous(memory_format=torch.channels_last)  # strided-but-dense parameters keep their memory order
        ps.append(t.requires_grad_(True))
    return ps


def test_hd_adamw_matches_torch_adamw_and_writes_shadows():
    from heltondetection_amd.optim import HdAdamW
    dev = torch.device("cuda:0")
    a, b = _params(dev, 0), [p.detach().clone().requires_grad_(True) for p in _params(dev, 0)]
    for pa, pb in zip(a, b):
        assert pa.stride() == pb.stride()
    shadows = {a[0]: torch.empty_like(a[0], dtype=torch.bfloat16), a[4]: torch.empty_like(a[4], dtype=torch.bfloat16)}
    groups = lambda ps: [{"params": [p for p in ps if p.ndim > 1], "weight_decay": 0.05},
                         {"params": [p for p in ps if p.ndim <= 1], "weight_decay": 0.0}]
    oa = HdAdamW(groups(a), lr=3e-3, betas=(0.937, 0.999), shadows=shadows)
    ob = torch.optim.AdamW(groups(b), lr=3e-3, betas=(0.937, 0.999), foreach=False, fused=False)
    sa = torch.optim.lr_scheduler.LambdaLR(oa, lambda s: 1.0 / (1 + s))
    sb = torch.optim.lr_scheduler.LambdaLR(ob, lambda s: 1.0 / (1 + s))
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(dev)
            pa.grad = torch.empty_like(pa).copy_(gr)   # same strides as the parameter, like dist.GradReducer's views
            pb.grad = torch.empty_like(pb).copy_(gr)
        oa.step(); ob.step(); sa.step(); sb.step()
        for pa, pb in zip(a, b):
            assert torch.allclose(pa, pb, rtol=2e-6, atol=2e-7), (step, pa.shape, float((pa - pb).abs().max()))
    for p in (a[0], a[4]):
        assert torch.equal(shadows[p], p.detach().to(torch.bfloat16))
    for pa, pb in zip(a, b):
        sa_, sb_ = oa.state[pa], ob.state[pb]
        assert float(sa_["step"]) == float(sb_["step"]) == 6.0
        assert torch.allclose(sa_["exp_avg"], sb_["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(sa_["exp_avg_sq"], sb_["exp_avg_sq"], rtol=1e-5, atol=1e-9)
    # the state dict is a torch.optim.AdamW state dict: it loads into one, and back
    oc = torch.optim.AdamW(groups([p.detach().clone().requires_grad_(True) for p in a]), lr=1.0, foreach=False, fused=False)
    oc.load_state_dict(oa.state_dict())
    oa.load_state_dict(oc.state_dict())
    for pa in a:
        pa.grad = torch.ones_like(pa)
    oa.step()  # the descriptor table is rebuilt for the loaded moment tensors
    assert all(float(oa.state[p]["step"]) == 7.0 for p in a)


def test_trainer_with_hd_adamw_tracks_torch_adamw():
    """Two YOLOv5s trainers from the same initialisation, one with the HIP optimizer and one with torch's, at lr = 0 the
    weights stay put and the moments agree; with lr > 0 both train (loss decreases)."""
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.train import Trainer
    dev = torch.device("cuda:0")
    g = syn.make_generator(0)
    img = syn.make_images(4, 128, g).to(dev).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = (t.to(dev) for t in syn.make_targets(4, 128, g, "coco", 80, 8))
    losses = {}
    for hd in (True, False):
        torch.manual_seed(0)
        model = build_detector(dict(type="YOLOv5", variant="s", num_classes=80, img_size=128)).to(dev)
        model = model.to(memory_format=torch.channels_last).train()
        tr = Trainer(model, lr=2e-3, total_steps=30, warmup_steps=2, device=dev, hd_optimizer=hd)
        assert type(tr.opt).__name__ == ("HdAdamW" if hd else "AdamW")
        ls = [float(tr.step(img, gtb, gtc, mv)[0]) for _ in range(12)]
        assert all(l == l for l in ls) and min(ls[-3:]) < ls[0], ls
        losses[hd] = ls
        if hd:  # the shadows the next forward reads are the bf16 cast of the masters the optimizer just wrote
            for master, shadow in tr.shadows.grad_sources().items():
                assert torch.equal(shadow.detach(), master.detach().to(torch.bfloat16))
    assert abs(losses[True][0] - losses[False][0]) < 1e-3 * abs(losses[False][0]) + 1e-4  # same first step


def test_hd_adamw_reads_bfloat16_gradients_from_the_shadows():
    """grad_from_shadows (the one-rank form of train.Trainer): a shadowed parameter whose own .grad is None takes its shadow's
    bfloat16 gradient where autograd left it; the update equals torch.optim.AdamW fed the same values as float32."""
    from heltondetection_amd.optim import HdAdamW
    dev = torch.device("cuda:0")
    a, b = _params(dev, 3), [p.detach().clone().requires_grad_(True) for p in _params(dev, 3)]
    shadows = {a[0]: torch.empty_like(a[0], dtype=torch.bfloat16).requires_grad_(True),
               a[4]: torch.empty_like(a[4], dtype=torch.bfloat16).requires_grad_(True),
               a[5]: torch.empty_like(a[5], dtype=torch.bfloat16).requires_grad_(True)}
    groups = lambda ps: [{"params": [p for p in ps if p.ndim > 1], "weight_decay": 0.05},
                         {"params": [p for p in ps if p.ndim <= 1], "weight_decay": 0.0}]
    oa = HdAdamW(groups(a), lr=3e-3, betas=(0.937, 0.999), shadows=shadows)
    oa.grad_from_shadows = True
    ob = torch.optim.AdamW(groups(b), lr=3e-3, betas=(0.937, 0.999), foreach=False, fused=False)
    g = torch.Generator().manual_seed(5)
    for step in range(5):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(dev)
            if pa in shadows:
                pa.grad = None
                shadows[pa].grad = torch.empty_like(shadows[pa]).copy_(gr)        # bf16, the parameter's memory order
                pb.grad = torch.empty_like(pb).copy_(shadows[pa].grad.float())    # the same values, float32
            else:
                pa.grad = torch.empty_like(pa).copy_(gr)
                pb.grad = torch.empty_like(pb).copy_(gr)
        oa.step(); ob.step()
        for pa, pb in zip(a, b):
            assert torch.allclose(pa, pb, rtol=2e-6, atol=2e-7), (step, pa.shape, float((pa - pb).abs().max()))
    for p, sh in shadows.items():
        assert torch.equal(sh.detach(), p.detach().to(torch.bfloat16))
    # without the switch a parameter without its own gradient is skipped, as in torch.optim
    oa.grad_from_shadows = False
    before = a[0].detach().clone()
    a[0].grad = None
    oa.step()
    assert torch.equal(a[0].detach(), before)
