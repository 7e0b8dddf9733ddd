"""Concatenations whose inputs are produced in place (ops.CatBuffer): the fused BN+act apply kernel and the SPPF
max-pool write straight into their channel slice of the concatenation buffer, and the backward kernels read the
matching slice of the gradient — instead of torch.cat's copy kernels (VERDICT r01 #7 / DESIGN.md open item).
Checked against the torch.cat path of the same modules: same values, same gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_act_and_pool_write_into_cat_slots(gpu, dtype):
    from heltondetection_amd import ops
    torch.manual_seed(0)
    B, H, W = 4, 20, 20
    ca, cb_ = 32, 64
    xa = _cl(torch.randn((B, ca, H, W), device=gpu).to(dtype)).requires_grad_(True)
    xb = _cl(torch.randn((B, cb_, H, W), device=gpu).to(dtype)).requires_grad_(True)
    wa, ba = torch.rand(ca, device=gpu, requires_grad=True), torch.randn(ca, device=gpu, requires_grad=True)
    wb, bb = torch.rand(cb_, device=gpu, requires_grad=True), torch.randn(cb_, device=gpu, requires_grad=True)
    g = _cl(torch.randn((B, ca + cb_ + cb_, H, W), device=gpu).to(dtype))

    def run(in_place):
        for t in (xa, xb, wa, ba, wb, bb):
            t.grad = None
        ops.bn_arena_reset(gpu)
        if in_place:
            buf = ops.CatBuffer(xa, [ca, cb_, cb_])
            ya = ops.bn_act(xa, wa, ba, None, None, 1e-3, 0.03, "silu", True, out=buf.slot(0))
            yb = ops.bn_act(xb, wb, bb, None, None, 1e-3, 0.03, "silu", True, out=buf.slot(1))
            assert buf.in_place(0, ya) and buf.in_place(1, yb)
            yp = ops.maxpool5(yb, out=buf.slot(2))          # strided input (slot 1) -> strided output (slot 2)
            z = buf.join([ya, yb, yp])
            assert z.data_ptr() == buf.buf.data_ptr() and z.is_contiguous(memory_format=torch.channels_last)
        else:
            ya = ops.bn_act(xa, wa, ba, None, None, 1e-3, 0.03, "silu", True)
            yb = ops.bn_act(xb, wb, bb, None, None, 1e-3, 0.03, "silu", True)
            z = torch.cat((ya, yb, ops.maxpool5(yb)), 1)
        z.backward(g)
        return z.detach().float().cpu(), [t.grad.detach().float().cpu() for t in (xa, xb, wa, ba, wb, bb)]

    z0, g0 = run(False)
    z1, g1 = run(True)
    assert torch.equal(z0, z1)                      # same kernels, same arithmetic: identical activations
    tol = 1e-5 if dtype == torch.float32 else 2e-2  # gradients: float atomics reorder the per-channel sums
    for a, b in zip(g0, g1):
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    # a foreign tensor (not produced in place) is copied into its slot by join, like torch.cat
    buf = ops.CatBuffer(xa, [ca, ca])
    other = _cl(torch.randn((B, ca, H, W), device=gpu).to(dtype))
    ya = ops.bn_act(xa, wa, ba, None, None, 1e-3, 0.03, "silu", True, out=buf.slot(0))
    z = buf.join([ya, other])
    assert torch.equal(z[:, ca:], other) and torch.equal(z[:, :ca], ya)


def test_yolov5_step_with_in_place_concatenation_matches_torch_cat(gpu):
    """Whole model, forward + loss + backward, fp32: the in-place concatenation path (C3, SPPF, PAFPN bottom-up) and
    the torch.cat path produce the same loss and the same gradients."""
    from heltondetection_amd import ops, synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.layers import FusedBNAct
    torch.manual_seed(0)
    model = build_detector(dict(type="YOLOv5", variant="s", num_classes=8, img_size=160)).to(gpu)
    model = model.to(memory_format=torch.channels_last).train()
    g = syn.make_generator(5)
    img = _cl(syn.make_images(4, 160, g).to(gpu))
    gtb, gtc, mv = [t.to(gpu) for t in syn.make_targets(4, 160, g, "coco", 8, 8)]
    out = {}
    for flag in (False, True):
        FusedBNAct.cat_in_place = flag
        try:
            model.zero_grad(set_to_none=True)
            ops.bn_arena_reset(gpu)
            loss, _ = model.forward_train(img, gtb, gtc, mv)
            loss.backward()
            out[flag] = (float(loss), [p.grad.detach().float().cpu().clone() for p in model.parameters()])
        finally:
            FusedBNAct.cat_in_place = True
    assert abs(out[True][0] - out[False][0]) <= 1e-5 * abs(out[False][0])
    num = sum(float(((a - b) ** 2).sum()) for a, b in zip(out[True][1], out[False][1]))
    den = sum(float((b ** 2).sum()) for b in out[False][1])
    assert (num / den) ** 0.5 < 1e-4, (num / den) ** 0.5
    # eval mode goes through hd_bn_act_infer with the same slots
    model.eval()
    with torch.no_grad():
        FusedBNAct.cat_in_place = False
        ref = [f.float().cpu() for f in model.extract(img)]
        FusedBNAct.cat_in_place = True
        got = [f.float().cpu() for f in model.extract(img)]
    for a, b in zip(got, ref):
        # not bit-equal: MIOpen may pick another solver for a convolution whose input is a fresh dense tensor rather
        # than a copied channel slice; the BN / pool kernels themselves are bit-identical (tests above and below)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_act_two_place_output_and_gradient(gpu, dtype):
    """hd_rows2: the fused BN writes channels [0, split) to a dense tensor and [split, C) into a concatenation slot
    (the merged cv1 | cv2 convolution of a C3 block), and its backward reads the two incoming gradients in place.
    Reference: the same kernels with one dense output, sliced by torch."""
    from heltondetection_amd import ops
    torch.manual_seed(1)
    B, H, W, C1, C2 = 3, 12, 20, 32, 32
    x = _cl(torch.randn((B, C1 + C2, H, W), device=gpu).to(dtype)).requires_grad_(True)
    w = torch.rand(C1 + C2, device=gpu, requires_grad=True)
    b = torch.randn(C1 + C2, device=gpu, requires_grad=True)
    ga = _cl(torch.randn((B, C1, H, W), device=gpu).to(dtype))
    gcat = _cl(torch.randn((B, 48 + C2, H, W), device=gpu).to(dtype))     # gradient of a [48 | C2] concatenation

    def run(split):
        for t in (x, w, b):
            t.grad = None
        ops.bn_arena_reset(gpu)
        if split:
            buf = ops.CatBuffer(x, [48, C2])
            y1 = torch.empty((B, C1, H, W), dtype=dtype, device=gpu, memory_format=torch.channels_last)
            a, c = ops.bn_act(x, w, b, None, None, 1e-3, 0.03, "silu", True, out=(y1, buf.slot(1)))
            assert a.data_ptr() == y1.data_ptr() and buf.in_place(1, c)
            torch.autograd.backward([a, c], [ga, gcat[:, 48:]])
        else:
            y = ops.bn_act(x, w, b, None, None, 1e-3, 0.03, "silu", True)
            a, c = y[:, :C1], y[:, C1:]
            torch.autograd.backward([a, c], [ga, gcat[:, 48:]])
        return a.detach().float().cpu(), c.detach().float().cpu(), [t.grad.detach().float().cpu() for t in (x, w, b)]

    a0, c0, g0 = run(False)
    a1, c1, g1 = run(True)
    assert torch.equal(a0, a1) and torch.equal(c0, c1)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for p, q in zip(g0, g1):
        assert float((p - q).abs().max()) <= tol * max(1.0, float(q.abs().max()))
    # inference (running statistics) with the same two places
    rm, rv = torch.randn(C1 + C2, device=gpu), torch.rand(C1 + C2, device=gpu) + 0.5
    with torch.no_grad():
        ref = ops.bn_act(x, w, b, rm, rv, 1e-3, 0.03, "silu", False)
        buf = ops.CatBuffer(x, [48, C2])
        y1 = torch.empty((B, C1, H, W), dtype=dtype, device=gpu, memory_format=torch.channels_last)
        a, c = ops.bn_act(x, w, b, rm, rv, 1e-3, 0.03, "silu", False, out=(y1, buf.slot(1)))
    assert torch.equal(a, ref[:, :C1]) and torch.equal(c, ref[:, C1:])


def test_c3_state_dict_round_trip_to_public_naming():
    from heltondetection_amd.layers import C3
    blk = C3(32, 64, n=2)
    sd = blk.state_dict()
    pub = C3.split_state_dict(sd, "")
    assert "cv1.conv.weight" in pub and "cv2.bn.running_var" in pub and not any(k.startswith("cv12.") for k in pub)
    assert pub["cv1.conv.weight"].shape == (32, 32, 1, 1)
    back = C3.merge_state_dict(pub, "")
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    blk.load_state_dict(back, strict=True)


@pytest.mark.parametrize("n,shortcut,act", [(1, False, "silu"), (2, True, "silu"), (1, False, "relu")])
def test_c2f_block_with_in_place_concatenation_matches_torch_cat(gpu, n, shortcut, act):
    """layers.C2f (the FasterRCNN PAFPN's fusion block, README.md:8): cv1 writes its two halves into two slots of the
    concatenation (one BN launch, an output in two places), every bottleneck writes its own slot — against the
    chunk / torch.cat path of the same module under bf16 autocast."""
    from heltondetection_amd import layers, ops
    torch.manual_seed(0)
    m = layers.C2f(128, 128, n, shortcut, act=act).to(gpu).to(memory_format=torch.channels_last).train()
    x0 = _cl(torch.randn((2, 128, 26, 26), device=gpu))
    gy = _cl(torch.randn((2, 128, 26, 26), device=gpu))
    out = {}
    for flag in (True, False):
        layers.FusedBNAct.cat_in_place = flag
        try:
            m.zero_grad(set_to_none=True)
            ops.bn_arena_reset(gpu)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * gy).sum().backward()
            ops.flush_wgrad()
            out[flag] = (y.detach().float(), x.grad.float(), [p.grad.detach().float().clone() for p in m.parameters()])
        finally:
            layers.FusedBNAct.cat_in_place = True
    ya, yb = out[True][0], out[False][0]
    assert float((ya - yb).abs().max()) <= 3e-2 * float(yb.abs().max())        # two bf16 pipelines, atomics reorder the sums
    assert float((out[True][1] - out[False][1]).abs().max()) <= 5e-2 * float(out[False][1].abs().max())
    for a, b in zip(out[True][2], out[False][2]):
        assert float((a - b).abs().max()) <= 5e-2 * float(b.abs().max()) + 1e-6
