"""hd_conv3x3s2_dgrad (csrc/conv3x3s2_dgrad.hip): the data gradient of the first stride-2 3x3 layer of the YOLOv5 backbone.
Compared with aten.convolution_backward of the same bf16-rounded operands in float32 (tolerance = one bf16 rounding of the result
plus float32 summation-order noise), exactly on small-integer data, and through the autograd node the model uses."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(dy, x_shape, w):
    x = torch.zeros(x_shape, dtype=torch.float32, device=dy.device)
    return torch.ops.aten.convolution_backward(dy.float(), x, w.float(), None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1,
                                               (True, False, False))[0]


def _mk(B, H, seed, dev, ints=False):
    g = torch.Generator().manual_seed(seed)
    if ints:
        dy = torch.randint(-3, 4, (B, 64, H, 160), generator=g).float()
        w = torch.randint(-2, 3, (64, 32, 3, 3), generator=g).float()
    else:
        dy = torch.randn((B, 64, H, 160), generator=g)
        w = torch.randn((64, 32, 3, 3), generator=g) * (9 * 64) ** -0.5
    dy = dy.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = w.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return dy, w


# (B, Hout): one row per workgroup walk / several; more images than workgroup slots; the bench shape
@pytest.mark.parametrize("B,H", [(1, 1), (1, 2), (2, 7), (3, 160), (16, 160), (300, 3)])
def test_stride2_dgrad_matches_fp32_reference(B, H):
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    dy, w = _mk(B, H, 11 + B + H, dev)
    assert ops.conv3x3s2_dgrad_supported(dy, (B, 32, 2 * H, 320), w)
    dx = ops.conv3x3s2_dgrad_rows(dy, w)
    ref = _ref(dy, (B, 32, 2 * H, 320), w)
    assert dx.shape == ref.shape and dx.dtype == torch.bfloat16 and dx.is_contiguous(memory_format=torch.channels_last)
    tol = 2.0 ** -8 * ref.abs() + 2e-5 * ref.abs().max()
    assert bool(((dx.float() - ref).abs() <= tol * 1.01).all()), float((dx.float() - ref).abs().max())


def test_stride2_dgrad_is_exact_on_integer_data_and_reads_channel_slices():
    """Small integers: every product and sum is exact in float32 and the result (|.| <= 9 * 64 * 6) exact in bf16 only up to 256,
    so the comparison is with the float32 reference ROUNDED to bf16.  dy is a channel slice of a wider tensor."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, H = 2, 9
    dy, w = _mk(B, H, 5, dev, ints=True)
    wide = torch.zeros((B, 160, H, 160), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    wide[:, 32:96] = dy
    view = wide[:, 32:96]
    dx = ops.conv3x3s2_dgrad_rows(view, w)
    ref = _ref(dy, (B, 32, 2 * H, 320), w).to(torch.bfloat16)
    assert torch.equal(dx, ref)


def test_stride2_dgrad_rejects_other_shapes():
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    lib = _lib.load()
    assert lib.hd_conv3x3s2_dgrad_supported(16, 160, 160, 32, 64) == 1
    assert lib.hd_conv3x3s2_dgrad_supported(16, 208, 208, 32, 64) == 0      # 832^2: stays with the library
    assert lib.hd_conv3x3s2_dgrad_supported(16, 160, 160, 64, 128) == 0
    dy, w = _mk(1, 4, 0, dev)
    assert not ops.conv3x3s2_dgrad_supported(dy, (1, 32, 7, 320), w)        # odd input height: 2 Hout - 1
    assert not ops.conv3x3s2_dgrad_supported(dy.float(), (1, 32, 8, 320), w)
    with pytest.raises(_lib.HdError):
        ops.conv3x3s2_dgrad_rows(dy[:, :, :, :80], w)


def test_first_stride2_layer_backward_runs_on_the_kernel_and_matches_the_library():
    """ops.conv3x3 (the autograd node ConvBNAct uses) with stride 2 on the layer's shape: dx from hd_conv3x3s2_dgrad, equal to the
    library's data gradient to bf16 rounding."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    x = torch.randn((2, 32, 48, 320), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn((64, 32, 3, 3), device=dev) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn((2, 64, 24, 160), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ops.enable_timing("conv3x3s2_dgrad")
    y = ops.conv3x3(x, w, 2)
    y.backward(gy)
    ops.flush_wgrad()
    torch.cuda.synchronize()
    ran = len(ops.collect_timing()["conv3x3s2_dgrad"])
    ops.disable_timing()
    assert ran == 1
    ref = _ref(gy, x.shape, w.detach())
    tol = 2.0 ** -8 * ref.abs() + 2e-5 * ref.abs().max()
    assert bool(((x.grad.float() - ref).abs() <= tol * 1.01).all())
    assert w.grad is not None and bool(torch.isfinite(w.grad.float()).all())


# ---- the same layer's forward: hd_conv3x3s2_fwd (csrc/conv3x3s2_fwd.hip) ---------------------------------------------------------
def _mkx(B, H, seed, dev, ints=False):
    g = torch.Generator().manual_seed(seed)
    if ints:
        x = torch.randint(-3, 4, (B, 32, 2 * H, 320), generator=g).float()
        w = torch.randint(-2, 3, (64, 32, 3, 3), generator=g).float()
    else:
        x = torch.randn((B, 32, 2 * H, 320), generator=g)
        w = torch.randn((64, 32, 3, 3), generator=g) * (9 * 32) ** -0.5
    x = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = w.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return x, w


@pytest.mark.parametrize("B,H", [(1, 1), (1, 2), (2, 7), (3, 160), (16, 160), (300, 3)])
def test_stride2_forward_matches_fp32_reference_and_statistics(B, H):
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    x, w = _mkx(B, H, 23 + B + H, dev)
    rep = _lib.load().hd_bn_act_replicas()
    stats = torch.zeros(rep * 2 * 64, dtype=torch.float32, device=dev)
    assert ops.conv3x3s2_fwd_supported(x, w)
    y = ops.conv3x3s2_fwd_rows(x, w, stats)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, 2, 1)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    tol = 2.0 ** -8 * ref.abs() + 2e-5 * ref.abs().max()
    assert bool(((y.float() - ref).abs() <= tol * 1.01).all()), float((y.float() - ref).abs().max())
    yf = y.double()   # the epilogue's statistics are those of the bf16 tensor it wrote
    want = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    got = stats.view(rep, 2, 64).double().sum(0)
    scale = torch.stack([yf.abs().sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]) + 1e-6
    assert float(((got - want).abs() / scale).max()) < 2e-6


def test_stride2_forward_is_exact_on_integer_data_and_reads_channel_slices():
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, H = 2, 9
    x, w = _mkx(B, H, 6, dev, ints=True)
    wide = torch.zeros((B, 96, 2 * H, 320), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    wide[:, 32:64] = x
    y = ops.conv3x3s2_fwd_rows(wide[:, 32:64], w)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, 2, 1).to(torch.bfloat16)
    assert torch.equal(y, ref)


def test_first_stride2_layer_module_forward_and_backward_match_the_library_path():
    """ConvBNAct(32, 64, 3, 2) on the layer's shape under bf16 autocast: hd_conv3x3s2_fwd (+ statistics), hd_conv3x3s2_dgrad and
    hd_conv3x3s2_wgrad_c32 against the all-library path of the same module."""
    from heltondetection_amd import layers, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = layers.ConvBNAct(32, 64, 3, 2).to(dev).to(memory_format=torch.channels_last).train()
    x0 = torch.randn((2, 32, 48, 320), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((2, 64, 24, 160), device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        layers.ConvBNAct.pointwise = on
        try:
            m.zero_grad(set_to_none=True)
            m.bn.running_mean.zero_(); m.bn.running_var.fill_(1.0)
            ops.bn_arena_reset(dev)
            x = x0.clone().requires_grad_(True)
            ops.enable_timing("conv3x3s2_fwd", "conv3x3s2_dgrad", "conv3x3s2_wgrad", "bn_stats_kernel")
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * gy).sum().backward()
            ops.flush_wgrad()
            torch.cuda.synchronize()
            timed = ops.collect_timing()
            ops.disable_timing()
            assert len(timed["conv3x3s2_fwd"]) == len(timed["conv3x3s2_dgrad"]) == len(timed["conv3x3s2_wgrad"]) == (1 if on else 0)
            assert len(timed["bn_stats_kernel"]) == (0 if on else 1)
            res[on] = (y.float().detach(), x.grad.float(), m.conv.weight.grad.float(), m.bn.weight.grad.float(),
                       m.bn.bias.grad.float(), m.bn.running_mean.clone(), m.bn.running_var.clone())
        finally:
            layers.ConvBNAct.pointwise = True
    for name, a, b in zip(("y", "dx", "dw", "dgamma", "dbeta", "running_mean", "running_var"), res[True], res[False]):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-6))
        assert err < 3e-2, (name, err)   # two bf16 pipelines with different accumulation orders


# ---- the same layer's weight gradient: hd_conv3x3s2_wgrad_c32 (csrc/conv3x3s2_wgrad.hip) -----------------------------------------------
def _wref(dy, x, w_shape):
    w = torch.zeros(w_shape, dtype=torch.float32, device=dy.device)
    return torch.ops.aten.convolution_backward(dy.float(), x.float(), w, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1,
                                               (False, True, False))[1]


@pytest.mark.parametrize("B,H", [(1, 1), (1, 2), (2, 7), (3, 160), (16, 160), (300, 3)])
def test_stride2_weight_gradient_matches_fp32_reference(B, H):
    """Float32 partials per workgroup, summed in float32, one bf16 rounding at the end: within one bf16 rounding of the float32
    reference plus the summation-order noise of a B x H x 160-term sum."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    x, _ = _mkx(B, H, 31 + B + H, dev)
    dy, _ = _mk(B, H, 37 + B + H, dev)
    assert ops.conv3x3s2_wgrad_supported(dy, x)
    dw = ops.conv3x3s2_wgrad_rows(dy, x)
    ref = _wref(dy, x, (64, 32, 3, 3))
    assert dw.shape == ref.shape and dw.dtype == torch.bfloat16 and dw.is_contiguous(memory_format=torch.channels_last)
    tol = 2.0 ** -8 * ref.abs() + 1e-4 * ref.abs().max()
    assert bool(((dw.float() - ref).abs() <= tol * 1.01).all()), float((dw.float() - ref).abs().max() / ref.abs().max())


def test_stride2_weight_gradient_is_exact_on_integer_data_and_reads_channel_slices():
    """Small integers on a 2 x 3-row problem: every product and partial sum is an integer below 2^24 (exact in float32); the result
    is compared with the float32 reference rounded to bf16.  Both operands are channel slices of wider tensors."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, H = 2, 3
    x, _ = _mkx(B, H, 8, dev, ints=True)
    dy, _ = _mk(B, H, 9, dev, ints=True)
    widex = torch.zeros((B, 96, 2 * H, 320), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    widex[:, 64:96] = x
    widey = torch.zeros((B, 128, H, 160), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    widey[:, 0:64] = dy
    dw = ops.conv3x3s2_wgrad_rows(widey[:, 0:64], widex[:, 64:96])
    ref = _wref(dy, x, (64, 32, 3, 3)).to(torch.bfloat16)
    assert torch.equal(dw, ref.contiguous(memory_format=torch.channels_last))
