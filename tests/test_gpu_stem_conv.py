"""csrc/stem_conv.hip: the YOLOv5 v6.x stem Conv2d(3, Cout, 6, 2, 2) — forward with the BatchNorm statistics in the epilogue,
weight gradient as row-sliced partials + slice sum.  Checked on exact small-integer data (every product and partial sum is
exactly representable: the im2col-free fragment addressing, the ring of input rows, borders, ragged widths and the statistics
must give the exact sums) and against a float32 convolution of the same bf16-rounded operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, H, W, Cout): the 640 / 832 / 1280 inputs, widths whose half is not a multiple of 32, heights that leave ragged row
# slices, a single short image
SHAPES = [(2, 64, 64, 32), (1, 40, 72, 32), (3, 36, 100, 64), (2, 128, 832, 32), (1, 24, 1280, 32), (16, 640, 640, 32),
          (1, 10, 8, 64), (2, 640, 640, 64)]


def _mk(B, H, W, N, seed, dev, dtype, ints):
    g = torch.Generator().manual_seed(seed)
    if ints:
        x = torch.randint(-3, 4, (B, 3, H, W), generator=g).float()
        w = torch.randint(-2, 3, (N, 3, 6, 6), generator=g).float()
    else:
        x = torch.rand((B, 3, H, W), generator=g)
        w = torch.randn((N, 3, 6, 6), generator=g) * 0.1
    x = x.to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = w.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return x, w


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_forward_and_statistics_exact_on_integer_data(shape, dtype):
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    B, H, W, N = shape
    x, w = _mk(B, H, W, N, H + W + N, dev, dtype, True)
    rep = _lib.load().hd_bn_act_replicas()
    stats = torch.zeros(rep * 2 * N, dtype=torch.float32, device=dev)
    y = ops.stem_conv_rows(x, w, stats)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, 2, 2)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    assert float(ref.abs().max()) <= 256.0            # exactly representable in bf16
    assert torch.equal(y.float(), ref)
    yd = ref.double()
    want = torch.stack([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))])
    got = stats.view(rep, 2, N).double().sum(0)
    assert float(((got - want).abs() / (want.abs() + 1.0)).max()) < 2e-6


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_weight_gradient_exact_on_integer_data(shape, dtype):
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, H, W, N = shape
    x, w = _mk(B, H, W, N, 3 * H + W + N, dev, dtype, True)
    g = torch.Generator().manual_seed(H * W)
    # sparse small gradients keep the exact sums below 2^24 at the largest shapes
    dy = (torch.randint(-2, 3, (B, N, H // 2, W // 2), generator=g) * (torch.rand((B, N, H // 2, W // 2), generator=g) < 0.02)).float()
    dy = dy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = ops.stem_conv_wgrad_rows(dy, x)
    ops.flush_wgrad()
    ref = torch.nn.grad.conv2d_weight(x.float(), (N, 3, 6, 6), dy.float(), stride=2, padding=2)
    assert dw.shape == ref.shape and dw.is_contiguous(memory_format=torch.channels_last)
    assert float(ref.abs().max()) < 2.0 ** 24
    assert torch.equal(dw.float(), ref.to(torch.bfloat16).float()), float((dw.float() - ref).abs().max())


@pytest.mark.parametrize("shape", [(4, 160, 160, 32), (2, 96, 200, 64)])
def test_stem_matches_fp32_convolution_on_real_data(shape):
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, H, W, N = shape
    x, w = _mk(B, H, W, N, 11, dev, torch.float32, False)
    y = ops.stem_conv_rows(x, w, None)
    xr = x.to(torch.bfloat16).float()   # the kernel rounds the pixels to bf16 on their way into LDS
    ref = torch.nn.functional.conv2d(xr, w.float(), None, 2, 2)
    tol = 2.0 ** -8 * ref.abs() + 1e-5 * ref.abs().max()
    assert bool(((y.float() - ref).abs() <= tol * 1.01).all()), float((y.float() - ref).abs().max())
    dy = torch.randn(ref.shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = ops.stem_conv_wgrad_rows(dy, x)
    ops.flush_wgrad()
    rw = torch.nn.grad.conv2d_weight(xr, (N, 3, 6, 6), dy.float(), stride=2, padding=2)
    tolw = 2.0 ** -8 * rw.abs() + 2e-5 * rw.abs().max()
    assert bool(((dw.float() - rw).abs() <= tolw * 1.01).all()), float(((dw.float() - rw).abs() / tolw).max())


def test_stem_layer_path_matches_the_library_path():
    """layers.ConvBNAct(3, 32, 6, 2, 2) under autocast with the stem kernels on vs MIOpen + the separate statistics kernel."""
    from heltondetection_amd import layers, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = layers.ConvBNAct(3, 32, 6, 2, 2).to(dev).to(memory_format=torch.channels_last).train()
    x = torch.rand((4, 3, 128, 128), device=dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((4, 32, 64, 64), device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        layers.ConvBNAct.stem = on
        try:
            m.zero_grad(set_to_none=True)
            m.bn.running_mean.zero_(); m.bn.running_var.fill_(1.0)
            ops.bn_arena_reset(dev)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * gy).sum().backward()
            ops.flush_wgrad()
            res[on] = (y.float().clone(), m.conv.weight.grad.float().clone(), m.bn.weight.grad.float().clone(),
                       m.bn.bias.grad.float().clone(), m.bn.running_mean.clone(), m.bn.running_var.clone())
        finally:
            layers.ConvBNAct.stem = True
    for name, a, b in zip(("y", "dw", "dgamma", "dbeta", "running_mean", "running_var"), res[True], res[False]):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-6))
        assert err < 3e-2, (name, err)   # two bf16 pipelines with different accumulation orders
