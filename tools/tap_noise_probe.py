"""A/A and A/B spread of FasterRCNN.extract (ResNet50 + PAFPN, bf16) features and gradients: forks on vs off vs rerun."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import layers, ops
from heltondetection_amd.detectors import build_detector
gpu = torch.device("cuda:0")
def run(model, img0, on):
    layers.ConvBNAct.fork_shortcut = on
    model.zero_grad(set_to_none=True)
    ops.bn_arena_reset(gpu)
    img = img0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats = model.extract(img)
    loss = sum((f.float() * torch.cos(torch.arange(f.numel(), device=gpu, dtype=torch.float32)).view_as(f)).sum() for f in feats)
    loss.backward(); ops.flush_wgrad()
    params = [p for p in list(model.backbone.parameters()) + list(model.neck.parameters()) if p.grad is not None]
    layers.ConvBNAct.fork_shortcut = True
    return [f.detach().float() for f in feats], [p.grad.float().clone() for p in params] + [img.grad.float().clone()]
def spread(a, b):
    return max(float((x - y).norm() / (y.norm() + 1e-12)) for x, y in zip(a, b))
for shape in [(2, 3, 128, 160), (4, 3, 256, 320)]:
    torch.manual_seed(3)
    model = build_detector(dict(type="FasterRCNN", num_classes=4)).to(gpu).to(memory_format=torch.channels_last).train()
    img0 = torch.randn(shape, device=gpu).contiguous(memory_format=torch.channels_last)
    r = {k: run(model, img0, on) for k, on in [("off1", False), ("off2", False), ("on1", True), ("on2", True)]}
    print(shape, "feats  off/off %.4f on/on %.4f on/off %.4f" % (spread(r["off1"][0], r["off2"][0]), spread(r["on1"][0], r["on2"][0]), spread(r["on1"][0], r["off1"][0])))
    print(shape, "grads  off/off %.4f on/on %.4f on/off %.4f" % (spread(r["off1"][1], r["off2"][1]), spread(r["on1"][1], r["on2"][1]), spread(r["on1"][1], r["off1"][1])), flush=True)
