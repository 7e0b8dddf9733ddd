"""The Detect convolution on the HIP kernels (csrc/pointwise_conv.hip: hd_head_conv_fwd, csrc/pointwise_bwd.hip:
hd_head_conv_bwd) and the loss kernels on its padded rows (csrc/yolo.hip, is_dense_nhwc kind 2).

Bounds, stated before the first run:
  forward   float32 accumulate of bf16 products + float32 bias, unrounded: against torch's float32 matmul of the same bf16
            values the difference is summation order only -> |diff| <= 1e-5 * (1 + |ref|) * sqrt(Cin / 32).
  backward  dx / dw are bf16 roundings of float32 sums: <= 2^-7 relative to the tensor's max (one bf16 ulp at the top of the range
            plus summation order), dw additionally scaled by sqrt(R) cancellation -> checked as rel-L2 <= 4e-3.
  loss      the padded layout changes addressing only; loss sums and gradients differ from the dense layout's by the order of
            float32 atomics alone (workgroup partials; cells matched more than once) -> <= 1e-6 relative, bias gradient 1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rows(B, C, H, W, dev, scale=1.0):
    return (torch.randn(B, C, H, W, device=dev) * scale).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("B,H,W,K,N", [
    (2, 20, 20, 128, 255), (2, 10, 10, 256, 255), (1, 5, 7, 512, 255), (2, 8, 8, 64, 75), (1, 9, 5, 32, 18),
    (1, 12, 12, 1024, 255), (3, 7, 3, 128, 33), (16, 80, 80, 128, 255), (4, 16, 16, 128, 39), (4, 4, 4, 512, 39),
])
def test_head_conv_forward_matches_float32_matmul(gpu, B, H, W, K, N):
    from heltondetection_amd import ops
    torch.manual_seed(B * 1000 + K + N)
    x = _rows(B, K, H, W, gpu)
    w = (torch.randn(N, K, device=gpu) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=gpu)
    P = ops.head_conv_pitch(N)
    assert P >= N and P in (32, 64, 128, 256)
    y = ops.head_conv_rows(x, w, b)
    assert y.shape == (B, P, H, W) and y.dtype == torch.float32 and y.is_contiguous(memory_format=torch.channels_last)
    ref = torch.einsum("bkhw,nk->bnhw", x.float(), w.float()) + b.view(1, N, 1, 1)
    tol = 1e-5 * (1 + ref.abs()) * (K / 32) ** 0.5
    assert bool(((y[:, :N] - ref).abs() <= tol).all()), float((y[:, :N] - ref).abs().max())
    assert float(y[:, N:].abs().max() if P > N else 0.0) == 0.0     # padding channels are zeros


@pytest.mark.parametrize("B,H,W,K,N", [
    (2, 20, 20, 128, 255), (2, 10, 10, 256, 255), (1, 5, 7, 512, 255), (2, 8, 8, 64, 75), (1, 9, 5, 32, 18),
    (2, 6, 6, 192, 255), (16, 40, 40, 256, 255), (4, 16, 16, 128, 39), (4, 8, 8, 256, 39), (4, 4, 4, 512, 39), (2, 8, 8, 128, 100),
])
def test_head_conv_backward_matches_torch(gpu, B, H, W, K, N):
    from heltondetection_amd import ops
    torch.manual_seed(B * 77 + K + N)
    x = _rows(B, K, H, W, gpu).requires_grad_(True)
    w = (torch.randn(N, K, 1, 1, device=gpu) / K ** 0.5).to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(N, device=gpu)
    assert ops.head_conv_supported(x, w)
    y = ops.head_conv(x, w, b)
    P = y.shape[1]
    gy = torch.zeros_like(y)
    gy[:, :N] = torch.randn(B, N, H, W, device=gpu) * 0.1            # padding channels of the gradient stay zero
    y.backward(gy)
    ops.flush_wgrad()
    g16 = gy[:, :N].to(torch.bfloat16).float()
    dx_ref = torch.einsum("bnhw,nk->bkhw", g16, w.detach().float().view(N, K))
    dw_ref = torch.einsum("bnhw,bkhw->nk", g16, x.detach().float())
    assert x.grad.dtype == torch.bfloat16 and w.grad.shape == w.shape
    ex = (x.grad.float() - dx_ref).abs().max() / dx_ref.abs().max()
    assert float(ex) <= 2.0 ** -7, float(ex)
    dw = w.grad.float().view(N, K)
    rel = (dw - dw_ref).norm() / dw_ref.norm()
    assert float(rel) <= 4e-3, float(rel)
    assert float((dw - dw_ref).abs().max() / dw_ref.abs().max()) <= 2.0 ** -7


def _loss_inputs(dev, B=4, nc=80, sizes=((16, 16), (8, 8), (4, 4)), M=6, seed=0):
    from heltondetection_amd import ops
    from heltondetection_amd.heads import YOLOV5_ANCHORS
    g = torch.Generator(device="cpu").manual_seed(seed)
    na, no = 3, nc + 5
    strides = (8.0, 16.0, 32.0)
    img = sizes[0][0] * 8
    cxy = torch.rand(B, M, 2, generator=g) * img
    wh = torch.rand(B, M, 2, generator=g) * img * 0.4 + 4
    boxes = torch.cat([cxy - wh / 2, cxy + wh / 2], -1).clamp(0, img).to(dev)
    cls = torch.randint(0, nc, (B, M), generator=g, dtype=torch.int32).to(dev)
    valid = torch.full((B,), M, dtype=torch.int32, device=dev)
    shapes = [(na, h, w, no) for h, w in sizes]
    slots = ops.yolo_assign(shapes, strides, YOLOV5_ANCHORS, boxes, cls, valid)
    dense = [torch.randn(B, h, w, na * no, generator=g).to(dev) for h, w in sizes]
    return strides, YOLOV5_ANCHORS, slots, M, dense, na, no


@pytest.mark.parametrize("nc", [80, 20, 1])
def test_yolo_loss_on_padded_rows_equals_dense_layout(gpu, nc):
    """The same logits in the dense [B,H,W,na*no] layout and in rows padded to head_conv_pitch: identical loss, identical
    gradient in the first na*no columns, zeros in the padding, identical bias gradient."""
    from heltondetection_amd import ops
    strides, anchors, slots, M, dense, na, no = _loss_inputs(gpu, nc=nc)
    N = na * no
    P = ops.head_conv_pitch(N)
    out = {}
    for padded in (False, True):
        preds, roots = [], []
        biases = [torch.zeros(N, device=gpu, requires_grad=True) for _ in dense]
        for d in dense:
            B, H, W, _ = d.shape
            if padded:
                r = torch.zeros(B, H, W, P, device=gpu)
                r[..., :N] = d
                r = r.permute(0, 3, 1, 2).requires_grad_(True)          # [B, P, H, W] channels-last
                roots.append(r)
                preds.append(r.as_strided((B, na, H, W, no), (H * W * P, no, W * P, P, 1)))
            else:
                r = d.clone().requires_grad_(True)
                roots.append(r)
                preds.append(r.view(B, H, W, na, no).permute(0, 3, 1, 2, 4))
        total, parts = ops.yolo_loss(preds, strides, anchors, slots, M, biases=biases, roots=roots if padded else None)
        total.backward()
        grads = [r.grad.permute(0, 2, 3, 1) if padded else r.grad for r in roots]
        out[padded] = (parts.clone(), grads, [b.grad.clone() for b in biases])
    assert torch.allclose(out[True][0], out[False][0], rtol=1e-6, atol=0.0)
    for gp, gd, bp, bd in zip(out[True][1], out[False][1], out[True][2], out[False][2]):
        assert float(gp[..., N:].abs().max()) == 0.0
        scale = float(gd.abs().max())
        assert float((gp[..., :N] - gd).abs().max()) <= 1e-6 * scale
        assert float((bp - bd).abs().max()) <= 1e-5 * float(bd.abs().max())


def test_yolov5_head_trains_through_head_conv_like_the_library_path(gpu):
    """YOLOv5Head.forward + loss in training mode with bf16 weight shadows, the HIP Detect convolution (padded rows) against
    the library convolution + cast (HD_HEAD_CONV off) on the same inputs.
      * loss: the library path rounds the logits to bf16 (relative 2^-9), ours does not -> <= 2e-3 relative;
      * our gradients against float32 matmuls of OUR OWN logit gradient (captured at the padded rows, rounded to bf16 as the
        kernel does): dx, dw <= 2^-7 of the tensor's max; bias gradient = column sums of the float32 logit gradient, 1e-4;
      * against the library path's gradients only a cosine >= 0.95 per tensor: d(CIoU)/d(box) is discontinuous where box
        edges cross, so the 2^-9 logit rounding of the library path flips single box gradients (measured: the 4 x 4 level
        with its handful of positives differs by 0.21 rel-L2, the two denser levels by 0.005)."""
    from heltondetection_amd import assigners, layers, ops  # noqa: F401  (assigners: registers YOLOv5Assigner)
    from heltondetection_amd.heads import YOLOv5Head
    torch.manual_seed(3)
    head = YOLOv5Head([128, 256, 512], num_classes=80).to(gpu).train()
    for m in head.m:
        m.weight_lp = m.weight.detach().to(torch.bfloat16).requires_grad_(True)
    B = 4
    feats = [_rows(B, c, s, s, gpu) for c, s in ((128, 16), (256, 8), (512, 4))]
    g = torch.Generator(device="cpu").manual_seed(1)
    cxy = torch.rand(B, 5, 2, generator=g) * 128
    wh = torch.rand(B, 5, 2, generator=g) * 50 + 4
    boxes = torch.cat([cxy - wh / 2, cxy + wh / 2], -1).clamp(0, 128).to(gpu)
    cls = torch.randint(0, 80, (B, 5), generator=g, dtype=torch.int32).to(gpu)
    valid = torch.full((B,), 5, dtype=torch.int32, device=gpu)
    out, root_grads = {}, []
    was_on, was, was16 = layers.LowpWeights.on, ops.HeadConvSwitch.enabled, ops.HeadConvSwitch.bf16_grad
    try:
        layers.LowpWeights.on = True
        ops.HeadConvSwitch.bf16_grad = False    # the float32 logit gradient flows through the padded rows, where the hooks sit
        for ours in (False, True):
            ops.HeadConvSwitch.enabled = ours
            fs = [f.clone().requires_grad_(True) for f in feats]
            for m in head.m:
                m.weight_lp.grad = None
                m.bias.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                preds = head(fs)
                assert (head._roots is not None) == ours
                if ours:
                    for r in head._roots:
                        r.register_hook(lambda g_: root_grads.append(g_.detach().clone()))
                total, parts = head.loss(preds, boxes, cls, valid)
            total.backward()
            ops.flush_wgrad()
            out[ours] = (float(total.detach()), [f.grad.float().clone() for f in fs],
                         [m.weight_lp.grad.float().clone() for m in head.m], [m.bias.grad.clone() for m in head.m])
    finally:
        layers.LowpWeights.on, ops.HeadConvSwitch.enabled, ops.HeadConvSwitch.bf16_grad = was_on, was, was16
    assert abs(out[True][0] - out[False][0]) <= 2e-3 * abs(out[False][0])
    assert len(root_grads) == 3
    by_hw = {tuple(g_.shape[2:]): g_ for g_ in root_grads}
    for l, (f, m) in enumerate(zip(feats, head.m)):
        gy = by_hw[tuple(f.shape[2:])]                                  # [B, 256, H, W] float32, padding channels zero
        assert float(gy[:, 255:].abs().max()) == 0.0
        g16 = gy[:, :255].to(torch.bfloat16).float()
        w = m.weight_lp.detach().float().view(255, -1)
        dx_ref = torch.einsum("bnhw,nk->bkhw", g16, w)
        dw_ref = torch.einsum("bnhw,bkhw->nk", g16, f.float())
        dx, dw, db = out[True][1][l], out[True][2][l].view(255, -1), out[True][3][l]
        assert float((dx - dx_ref).abs().max()) <= 2.0 ** -7 * float(dx_ref.abs().max())
        assert float((dw - dw_ref).abs().max()) <= 2.0 ** -7 * float(dw_ref.abs().max())
        db_ref = gy[:, :255].sum((0, 2, 3))
        assert float((db - db_ref).abs().max()) <= 1e-4 * float(db_ref.abs().max())
    for k in (1, 2, 3):
        for a, b in zip(out[True][k], out[False][k]):
            cos = float((a * b).sum() / (a.norm() * b.norm()))
            assert cos >= 0.95, (k, cos)


def test_bf16_logit_gradient_rows_match_the_float32_gradient_path(gpu):
    """ops.yolo_loss(heads=...) — the loss backward writes bf16 padded rows and runs hd_head_conv_bwd itself — against
    ops.yolo_loss(roots=...) (float32 rows, cast once by ops.head_conv's backward) on the same logits.  The two round the same
    float32 value once, except in cells matched by several targets, whose box / class terms accumulate in bf16 (one extra
    rounding, 2^-9 relative, per addition).  Bounds: same loss (same forward kernels, sums differ by atomic order: 1e-6); dx, dw
    within 2^-7 of the tensor's max and 2e-3 rel-L2; bias gradients (float32 sums of unrounded terms in both) 1e-5."""
    from heltondetection_amd import ops
    from heltondetection_amd.heads import YOLOV5_ANCHORS
    torch.manual_seed(11)
    B, nc, na = 8, 80, 3
    no, N = nc + 5, na * (nc + 5)
    strides, sizes, chans = (8.0, 16.0, 32.0), ((16, 16), (8, 8), (4, 4)), (128, 256, 512)
    g = torch.Generator(device="cpu").manual_seed(5)
    M = 12                                                   # many targets on a 128 px image: cells matched more than once
    cxy = torch.rand(B, M, 2, generator=g) * 128
    wh = torch.rand(B, M, 2, generator=g) * 40 + 6
    boxes = torch.cat([cxy - wh / 2, cxy + wh / 2], -1).clamp(0, 128).to(gpu)
    cls = torch.randint(0, nc, (B, M), generator=g, dtype=torch.int32).to(gpu)
    valid = torch.full((B,), M, dtype=torch.int32, device=gpu)
    slots = ops.yolo_assign([(na, h, w, no) for h, w in sizes], strides, YOLOV5_ANCHORS, boxes, cls, valid)
    assert int((slots[0] >= 0).sum()) > 0
    feats = [_rows(B, c, h, w, gpu) for c, (h, w) in zip(chans, sizes)]
    ws = [(torch.randn(N, c, 1, 1, device=gpu) / c ** 0.5).to(torch.bfloat16) for c in chans]
    bs = [torch.randn(N, device=gpu) * 0.1 - 3.0 for _ in chans]
    out = {}
    for lowp in (False, True):
        xs = [f.clone().requires_grad_(True) for f in feats]
        wl = [w.clone().requires_grad_(True) for w in ws]
        bl = [b.clone().requires_grad_(True) for b in bs]
        roots, preds = [], []
        for x, w, b in zip(xs, wl, bl):
            y = ops.head_conv(x, w, b)
            Bn, P, H, W = y.shape
            roots.append(y)
            preds.append(y.as_strided((Bn, na, H, W, no), (H * W * P, no, W * P, P, 1)))
        kw = dict(heads=list(zip(xs, wl))) if lowp else dict(roots=roots)
        total, parts = ops.yolo_loss(preds, strides, YOLOV5_ANCHORS, slots, M, biases=bl, **kw)
        total.backward()
        ops.flush_wgrad()
        out[lowp] = (parts.clone(), [x.grad.float() for x in xs], [w.grad.float() for w in wl], [b.grad.clone() for b in bl])
    assert torch.allclose(out[True][0], out[False][0], rtol=1e-6, atol=0.0)
    for k in (1, 2):
        for a, b in zip(out[True][k], out[False][k]):
            assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
            assert float((a - b).norm() / b.norm()) <= 2e-3
    for a, b in zip(out[True][3], out[False][3]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


def test_head_conv_takes_a_channel_slice_as_input(gpu):
    """x as a channel slice of a wider channels-last tensor (row stride 2 Cin — a concatenation's slot): forward and backward
    must equal the dense copy's bit for bit (same kernels, another row stride)."""
    from heltondetection_amd import ops
    torch.manual_seed(4)
    B, K, H, W, N = 2, 128, 9, 11, 255
    wide = _rows(B, 2 * K, H, W, gpu)
    xs = wide[:, K:]
    xd = xs.contiguous(memory_format=torch.channels_last)
    assert xs.stride(3) == 2 * K and not xs.is_contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, K, 1, 1, device=gpu) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=gpu)
    gy = torch.zeros((B, ops.head_conv_pitch(N), H, W), device=gpu).contiguous(memory_format=torch.channels_last)
    gy[:, :N] = torch.randn(B, N, H, W, device=gpu)
    out = []
    for x in (xs, xd):
        xx = x.detach().requires_grad_(True)
        ww = w.clone().requires_grad_(True)
        assert ops.head_conv_supported(xx, ww)
        y = ops.head_conv(xx, ww, b)
        y.backward(gy)
        ops.flush_wgrad()
        out.append((y.detach().clone(), xx.grad.clone(), ww.grad.clone()))
    for a, c in zip(out[0], out[1]):
        assert torch.equal(a, c)
