"""Bias + activation of a biased convolution (csrc/bn_act.hip: hd_bias_act_fwd / _bwd) and the RPN head built on it: the
reference's FasterRCNN head is conv3x3 + bias -> ReLU -> {objectness, deltas} 1x1 convolutions (README.md:8)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("act", [None, "relu"])
@pytest.mark.parametrize("shape", [(2, 256, 13, 11), (1, 64, 7, 5), (3, 24, 9, 9), (2, 8, 31, 3)])
def test_bias_act_is_one_rounding_of_the_float32_expression_and_its_backward_is_exact(shape, act, dtype):
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape))
    x = _cl(torch.randn(shape, generator=g).to(dev).to(dtype))
    b = (torch.randn(shape[1], generator=g) * 0.5).to(dev)
    dy = _cl(torch.randn(shape, generator=g).to(dev).to(dtype))
    xr = x.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y = ops.bias_act(xr, br, act)
    want = x.float() + b[None, :, None, None]
    want = (want.relu() if act == "relu" else want).to(dtype)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, want)
    y.backward(dy)
    dx_want = torch.where(want > 0, dy, torch.zeros_like(dy)) if act == "relu" else dy
    assert torch.equal(xr.grad, dx_want)
    db_want = dx_want.double().sum((0, 2, 3))
    scale = dx_want.double().abs().sum((0, 2, 3)) + 1e-9
    assert float(((br.grad.double() - db_want).abs() / scale).max()) < 1e-6
    # in place over a non-leaf (what layers.conv_bias_act does with the convolution's output)
    x2 = (x.clone().requires_grad_(True))
    t = x2 * 1.0
    y2 = ops.bias_act(t, b, act, inplace=True)
    assert y2.data_ptr() == t.data_ptr() and torch.equal(y2, want)
    y2.backward(dy)
    assert torch.equal(x2.grad, dx_want)


def test_bias_act_rejects_what_it_cannot_take():
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    x = _cl(torch.randn((1, 12, 4, 4), device=dev).bfloat16())  # 12 channels: not a multiple of the 16-byte vector
    assert not ops.bias_act_supported(x, "relu")
    with pytest.raises(_lib.HdError):
        ops.bias_act(x, torch.zeros(12, device=dev), "relu")
    x = _cl(torch.randn((1, 16, 4, 4), device=dev).bfloat16())
    assert not ops.bias_act_supported(x, "silu")  # the backward works from the output: ReLU or nothing
    with pytest.raises(_lib.HdError):
        ops.bias_act(x, torch.zeros(8, device=dev), "relu")


def test_rpn_head_matches_the_three_separate_biased_convolutions():
    """RPNHead.forward (bias-free 3x3 + fused bias/ReLU, objectness and deltas from one stacked 1x1) against the textbook
    form relu(conv(f)) -> obj(t), reg(t): forward values and every parameter / input gradient within bf16 tolerance."""
    import torch.nn.functional as F
    import heltondetection_amd.assigners  # noqa: F401  (registers MaxIoUAssigner)
    from heltondetection_amd.heads import RPNHead
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    head = RPNHead(in_channels=64, strides=(8, 16)).to(dev).to(memory_format=torch.channels_last)
    for m in (head.conv, head.obj, head.reg):
        torch.nn.init.normal_(m.weight, std=0.05)
        torch.nn.init.normal_(m.bias, std=0.1)
    feats0 = [_cl(torch.randn((2, 64, 20, 24), device=dev)), _cl(torch.randn((2, 64, 10, 12), device=dev))]

    def run(fused):
        head.zero_grad(set_to_none=True)
        feats = [f.clone().requires_grad_(True) for f in feats0]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if fused:
                lg, dl = head(feats)
            else:
                lgs, dls = [], []
                for f in feats:
                    t = F.relu(head.conv(f))
                    B = t.shape[0]
                    lgs.append(head.obj(t).float().permute(0, 2, 3, 1).reshape(B, -1))
                    dls.append(head.reg(t).float().permute(0, 2, 3, 1).reshape(B, -1, 4))
                lg, dl = torch.cat(lgs, 1), torch.cat(dls, 1)
        gl = torch.sin(torch.arange(lg.numel(), device=dev, dtype=torch.float32)).view_as(lg)
        gd = torch.cos(torch.arange(dl.numel(), device=dev, dtype=torch.float32)).view_as(dl)
        ((lg * gl).sum() + (dl * gd).sum()).backward()
        grads = [p.grad.float().clone() for p in head.parameters()] + [f.grad.float().clone() for f in feats]
        return lg.detach(), dl.detach(), grads

    lg1, dl1, g1 = run(True)
    lg0, dl0, g0 = run(False)
    assert lg1.shape == lg0.shape and dl1.shape == dl0.shape
    # two bf16 pipelines: the bias is added after the convolution's own rounding here and inside the library there, so hidden
    # values differ by one bf16 ulp and a few ReLU masks flip near zero: bounded in the mean (rel. L2) and in the maximum
    for a, b in [(lg1, lg0), (dl1, dl0)] + list(zip(g1, g0)):
        l2 = float((a - b).norm() / (b.norm() + 1e-12))
        mx = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert l2 < 1e-2 and mx < 2.0 ** -4, (tuple(a.shape), l2, mx)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 64, 16, 24), (1, 8, 7, 9), (3, 24, 13, 1), (1, 16, 1, 5), (2, 64, 52, 52), (1, 8, 2, 2)])
def test_maxpool3s2_equals_the_library_pool_forward_and_backward(shape, dtype):
    """hd_maxpool3s2_fwd / _bwd (the ResNet stem's pooling): values and gradients equal F.max_pool2d(x, 3, 2, 1) exactly, odd
    and even sizes.  The input takes nine distinct small integers arranged so that every 3x3 window sees each at most once
    (value = perm[b, c][(h % 3) * 3 + w % 3]): the argmax is unique, so the two implementations must agree bit for bit."""
    import torch.nn.functional as F
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, C, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    perm = torch.stack([torch.randperm(9, generator=g) for _ in range(B * C)]).view(B, C, 9).float()
    cell = (torch.arange(H).view(H, 1) % 3) * 3 + (torch.arange(W).view(1, W) % 3)
    x0 = perm[:, :, cell.reshape(-1)].view(B, C, H, W) - 4.0
    x = _cl(x0.to(dev).to(dtype)).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y = ops.maxpool3s2(x)
    ref = F.max_pool2d(xr, 3, 2, 1)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last) and torch.equal(y, ref)
    gy = _cl(torch.randint(-3, 4, ref.shape, generator=g).float().to(dev).to(dtype))
    y.backward(gy)
    ref.backward(gy)
    assert torch.equal(x.grad, xr.grad)
