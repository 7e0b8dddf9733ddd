"""hd_pointwise_conv (csrc/pointwise_conv.hip): the 1x1 convolution as an MFMA GEMM with the fused BatchNorm's
statistics in its epilogue.  Checked against float32 matmuls of the same bf16-rounded operands (tolerance = one bf16
rounding of the result, 2^-8 relative, plus the float32 accumulation-order noise)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(B, K, H, W, N, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, K, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((N, K, 1, 1), generator=g) * K ** -0.5).to(dev).to(torch.bfloat16)
    return x, w


def _ref(x, w):
    return torch.nn.functional.conv2d(x.float(), w.float())


# (B, Cin, H, W, Cout): every streaming-form tile shape (K 32/64/128/256 x column tiles 32/64/128), the K-loop form
# (K 96, 768), the long-reduction form of csrc/pointwise_longk.hip (K 512 with 128- and 64-wide column tiles, K 1024; three
# column tiles; more row blocks than workgroups), ragged row counts (not a multiple of 32 / 64 / 128) and a single-row input
SHAPES = [(2, 32, 20, 20, 32), (2, 64, 24, 20, 64), (1, 64, 17, 13, 128), (2, 128, 9, 11, 128), (1, 128, 16, 16, 64),
          (2, 256, 10, 10, 256), (1, 256, 7, 9, 128), (1, 256, 5, 5, 32), (2, 64, 16, 16, 256), (1, 32, 33, 7, 96),
          (2, 512, 10, 10, 512), (1, 1024, 6, 6, 512), (1, 96, 7, 7, 160), (1, 64, 1, 1, 64), (16, 64, 80, 80, 64),
          # more than 8 column tiles per launch (ResNet50's 256 -> 1024; 12 and 20 tiles: not powers of two)
          (2, 256, 13, 13, 1024), (1, 256, 9, 9, 768), (1, 64, 31, 9, 640), (4, 256, 52, 52, 1024),
          (16, 512, 20, 20, 256), (16, 512, 20, 20, 512), (4, 512, 23, 17, 384), (16, 1024, 20, 20, 512), (16, 512, 40, 40, 256),
          (3, 1024, 11, 9, 192), (1, 512, 1, 1, 128), (20, 512, 40, 40, 128), (1, 768, 9, 7, 128),
          (16, 512, 32, 32, 1024), (9, 512, 37, 41, 640)]   # 16 384 / 13 653 rows x 8 / 5 column tiles: four row blocks per workgroup


@pytest.mark.parametrize("shape", SHAPES)
def test_pointwise_conv_matches_fp32_reference_and_statistics(shape):
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    x, w = _mk(B, K, H, W, N, 7 + K + N, dev)
    rep = _lib.load().hd_bn_act_replicas()
    stats = torch.zeros(rep * 2 * N, dtype=torch.float32, device=dev)
    y = ops.pointwise_conv_rows(x, w.view(N, K), stats)
    ref = _ref(x, w)
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    tol = 2.0 ** -8 * ref.abs() + 1e-5 * ref.abs().max()
    assert bool(((y.float() - ref).abs() <= tol * 1.01).all()), float((y.float() - ref).abs().max())
    # the epilogue's statistics are those of the bf16 tensor it wrote
    yf = y.double()
    want = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    got = stats.view(rep, 2, N).double().sum(0)
    scale = torch.stack([yf.abs().sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))]) + 1e-6
    assert float(((got - want).abs() / scale).max()) < 2e-6


def test_pointwise_conv_reads_and_writes_channel_slices():
    """x a channel slice of a wider channels-last tensor, y written into a slice of a concatenation buffer."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, H, W = 2, 12, 10
    g = torch.Generator().manual_seed(3)
    wide = torch.randn((B, 192, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = wide[:, 64:128]
    w = (torch.randn((96, 64, 1, 1), generator=g) * 0.125).to(dev).to(torch.bfloat16)
    buf = torch.full((B, 160, H, W), 7.0, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    out = buf[:, 32:128]
    ops.pointwise_conv_rows(x, w.view(96, 64), None, out)
    ref = _ref(x, w)
    assert torch.allclose(out.float(), ref, rtol=2.0 ** -7, atol=1e-3)
    assert bool((buf[:, :32] == 7.0).all()) and bool((buf[:, 128:] == 7.0).all())  # neighbours untouched


def test_long_reduction_form_reads_and_writes_channel_slices():
    """As above for csrc/pointwise_longk.hip: a 512-channel slice of a 768-channel concatenation in, a slice of a wider buffer
    out, 2 x 13 x 11 = 286 rows (three row blocks, the last ragged), statistics of the written tensor."""
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    B, H, W = 2, 13, 11
    g = torch.Generator().manual_seed(5)
    wide = torch.randn((B, 768, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = wide[:, 128:640]
    w = (torch.randn((256, 512, 1, 1), generator=g) * 512 ** -0.5).to(dev).to(torch.bfloat16)
    assert _lib.load().hd_pointwise_conv_resident(B * H * W, 512, 256) == 1
    buf = torch.full((B, 384, H, W), 7.0, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    out = buf[:, 64:320]
    rep = _lib.load().hd_bn_act_replicas()
    stats = torch.zeros(rep * 2 * 256, dtype=torch.float32, device=dev)
    ops.pointwise_conv_rows(x, w.view(256, 512), stats, out)
    ref = _ref(x, w)
    tol = 2.0 ** -8 * ref.abs() + 1e-5 * ref.abs().max()
    assert bool(((out.float() - ref).abs() <= tol * 1.01).all())
    assert bool((buf[:, :64] == 7.0).all()) and bool((buf[:, 320:] == 7.0).all())  # neighbours untouched
    yf = out.double()
    got = stats.view(rep, 2, 256).double().sum(0)
    assert torch.allclose(got[0], yf.sum((0, 2, 3)), rtol=0, atol=2e-6 * float(yf.abs().sum((0, 2, 3)).max()))
    assert torch.allclose(got[1], (yf * yf).sum((0, 2, 3)), rtol=2e-6, atol=0)


def test_pointwise_conv_rejects_what_it_cannot_take():
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    x, w = _mk(1, 48, 4, 4, 64, 0, dev)  # Cin not a multiple of 32
    assert not ops.pointwise_conv_supported(x, w)
    with pytest.raises(_lib.HdError):
        ops.pointwise_conv_rows(x, w.view(64, 48))
    x, w = _mk(1, 64, 4, 4, 64, 0, dev)
    assert not ops.pointwise_conv_supported(x.float(), w)  # bf16 only
    assert not ops.pointwise_conv_supported(x.contiguous(), w)  # NCHW rows are not channels-last rows


@pytest.mark.parametrize("cin,cout,hw", [(64, 64, 24), (128, 256, 10), (256, 128, 12), (512, 256, 6)])
def test_conv_bn_act_pointwise_path_matches_library_path(cin, cout, hw):
    """ConvBNAct forward + backward with the MFMA 1x1 path on vs MIOpen + the separate statistics kernel."""
    from heltondetection_amd import layers, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = layers.ConvBNAct(cin, cout, 1, 1).to(dev).to(memory_format=torch.channels_last).train()
    x0 = torch.randn((4, cin, hw, hw), device=dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((4, cout, hw, hw), device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        layers.ConvBNAct.pointwise = on
        try:
            m.zero_grad(set_to_none=True)
            m.bn.running_mean.zero_(); m.bn.running_var.fill_(1.0)
            ops.bn_arena_reset(dev)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * gy).sum().backward()
            res[on] = (y.float(), x.grad.float(), m.conv.weight.grad.float(), m.bn.weight.grad.float(),
                       m.bn.bias.grad.float(), m.bn.running_mean.clone(), m.bn.running_var.clone())
        finally:
            layers.ConvBNAct.pointwise = True
    names = ("y", "dx", "dw", "dgamma", "dbeta", "running_mean", "running_var")
    for name, a, b in zip(names, res[True], res[False]):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-6))
        assert err < 3e-2, (name, err)  # two bf16 pipelines with different accumulation orders


@pytest.mark.parametrize("shape", [(2, 64, 8, 8, 64), (2, 128, 9, 7, 64), (1, 64, 13, 5, 128), (2, 128, 8, 8, 128),
                                   (1, 256, 5, 5, 128), (1, 64, 13, 13, 192), (2, 512, 6, 6, 256), (16, 64, 40, 40, 64),
                                   # 32 x 32 tiles (a channel count that is not a multiple of 64)
                                   (2, 32, 20, 20, 32), (1, 96, 7, 9, 32), (2, 32, 16, 16, 64), (1, 160, 5, 5, 96), (4, 32, 160, 160, 32)])
def test_pointwise_wgrad_is_exact_on_integer_data(shape):
    """Small-integer operands: every product and partial sum is exactly representable in float32, so the transposed
    LDS fragments, the row slicing (ragged last slice, rows past the end zeroed) and the slab reduction must give the
    exact sums (then rounded once to bf16)."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    g = torch.Generator().manual_seed(K + N + H)
    x = torch.randint(-3, 4, (B, K, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randint(-3, 4, (B, N, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = ops.pointwise_wgrad_rows(dy, x)
    ref = torch.einsum("bnhw,bkhw->nk", dy.float(), x.float())
    assert dw.shape == (N, K) and dw.dtype == torch.bfloat16
    assert torch.equal(dw, ref.to(torch.bfloat16))


def test_pointwise_wgrad_reads_channel_slices():
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    wide_x = torch.randn((2, 192, 9, 9), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wide_dy = torch.randn((2, 256, 9, 9), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x, dy = wide_x[:, 64:128], wide_dy[:, 128:256]
    dw = ops.pointwise_wgrad_rows(dy, x).float()
    ref = torch.einsum("bnhw,bkhw->nk", dy.float(), x.float())
    assert torch.allclose(dw, ref, rtol=2.0 ** -7, atol=2e-2)


@pytest.mark.parametrize("shape", [(2, 32, 9, 9, 32), (2, 64, 8, 8, 64), (1, 128, 13, 5, 64), (2, 64, 7, 7, 128),
                                   (2, 128, 8, 8, 128), (1, 512, 5, 5, 256), (1, 256, 6, 6, 256), (1, 96, 9, 9, 32),
                                   (16, 128, 40, 40, 128)])
def test_pointwise_dgrad_from_the_forward_weight(shape):
    """dx = dy . w read through LDS transpose fragments: exact on small-integer data, and equal (bit for bit) to the
    same product computed by hd_pointwise_conv from a transposed copy of w."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape  # K = Cin, N = Cout
    g = torch.Generator().manual_seed(K * 3 + N)
    dy = torch.randint(-3, 4, (B, N, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randint(-2, 3, (N, K), generator=g).float().to(dev).to(torch.bfloat16)
    dx = ops.pointwise_dgrad_rows(dy, w)
    ref = torch.einsum("bnhw,nk->bkhw", dy.float(), w.float())
    assert dx.shape == ref.shape and torch.equal(dx.float(), ref.to(torch.bfloat16).float())
    dy = torch.randn(dy.shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((N, K), generator=g) * N ** -0.5).to(dev).to(torch.bfloat16)
    assert torch.equal(ops.pointwise_dgrad_rows(dy, w), ops.pointwise_conv_rows(dy, w.t().contiguous()))


def test_long_reduction_layer_runs_on_our_forward_and_backward():
    """Cin = 512 > 256: forward through hd_pointwise_conv's long-reduction form (csrc/pointwise_longk.hip, statistics in its
    epilogue; the library + a separate statistics launch until round 5), weight gradient through hd_pointwise_wgrad; results
    must match the all-library path."""
    from heltondetection_amd import layers, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = layers.ConvBNAct(512, 512, 1, 1).to(dev).to(memory_format=torch.channels_last).train()
    x0 = torch.randn((4, 512, 10, 10), device=dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((4, 512, 10, 10), device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        layers.ConvBNAct.pointwise = on
        try:
            m.zero_grad(set_to_none=True)
            ops.bn_arena_reset(dev)
            x = x0.clone().requires_grad_(True)
            ops.enable_timing("pointwise_wgrad", "pointwise_conv", "bn_stats_kernel")
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * gy).sum().backward()
            torch.cuda.synchronize()
            timed = ops.collect_timing()
            ops.disable_timing()
            assert len(timed["pointwise_wgrad"]) == (1 if on else 0)
            assert len(timed["pointwise_conv"]) == (1 if on else 0)
            assert len(timed["bn_stats_kernel"]) == (0 if on else 1)   # the statistics come from the convolution's epilogue
            res[on] = (y.float().detach(), x.grad.float(), m.conv.weight.grad.float())
        finally:
            layers.ConvBNAct.pointwise = True
    for name, a, b in zip(("y", "dx", "dw"), res[True], res[False]):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-6))
        assert err < 3e-2, (name, err)


@pytest.mark.parametrize("cin,cout", [(256, 1024), (1024, 256), (2048, 512), (64, 256)])
def test_resnet_width_layers_train_through_conv_bn_act(cin, cout):
    """ResNet50 bottleneck widths (up to 2048): whichever of our kernels does not take a shape, the layer must still
    train (library fallback per direction) and agree with the all-library path."""
    from heltondetection_amd import _lib, layers, ops
    dev = torch.device("cuda:0")
    lib = _lib.load()
    assert lib.hd_pointwise_conv_dgrad_supported(1000, 2048, 256) == 0  # wider than the kernels' 1024 columns
    assert lib.hd_pointwise_conv_resident(1000, 256, 2048) == 0
    assert lib.hd_pointwise_conv_dgrad_supported(1000, 1024, 256) == 1  # 16 column tiles of 64 (<= 32)
    assert lib.hd_pointwise_conv_resident(1000, 256, 1024) == 1
    torch.manual_seed(2)
    m = layers.ConvBNAct(cin, cout, 1, 1, act="relu").to(dev).to(memory_format=torch.channels_last).train()
    x0 = torch.randn((2, cin, 13, 13), device=dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((2, cout, 13, 13), device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        layers.ConvBNAct.pointwise = on
        try:
            m.zero_grad(set_to_none=True)
            ops.bn_arena_reset(dev)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * gy).sum().backward()
            res[on] = (y.float().detach(), x.grad.float(), m.conv.weight.grad.float())
        finally:
            layers.ConvBNAct.pointwise = True
    for name, a, b in zip(("y", "dx", "dw"), res[True], res[False]):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-6))
        assert err < 3e-2, (name, err)


# ---- conv1x1 + BatchNorm + activation as one autograd node: BatchNorm backward folded into the gradient kernels ----------
def _layer_run(m, x0, gy, res_in, out_pair, dev):
    """One forward + backward of a ConvBNAct layer under bf16 autocast -> the tensors a parity check looks at."""
    from heltondetection_amd import ops
    m.zero_grad(set_to_none=True)
    m.bn.running_mean.zero_(); m.bn.running_var.fill_(1.0)
    ops.bn_arena_reset(dev)
    x = x0.clone().requires_grad_(True)
    r = res_in.clone().requires_grad_(True) if res_in is not None else None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x, r, out_pair)
    if isinstance(y, tuple):
        loss = sum((t.float() * g).sum() for t, g in zip(y, gy))
        y = torch.cat([t.float() for t in y], 1)
    else:
        loss = (y.float() * gy).sum()
    loss.backward()
    ops.flush_wgrad()
    return {"y": y.float().clone(), "dx": x.grad.float().clone(), "dw": m.conv.weight.grad.float().clone(),
            "dgamma": m.bn.weight.grad.float().clone(), "dbeta": m.bn.bias.grad.float().clone(),
            "dres": r.grad.float().clone() if r is not None else None,
            "running_mean": m.bn.running_mean.clone(), "running_var": m.bn.running_var.clone()}


FUSED_CASES = [(64, 64, 24, "silu", False, False), (128, 128, 13, "silu", True, False), (128, 256, 10, "silu", False, True),
               (256, 128, 12, "relu", False, False), (256, 256, 9, None, False, False), (512, 256, 6, "silu", False, False),
               (64, 128, 40, "silu", False, True), (1024, 512, 5, "silu", False, False),
               (32, 32, 24, "silu", True, False), (32, 96, 11, "relu", False, False)]   # 32 x 32 weight-gradient tiles


@pytest.mark.parametrize("cin,cout,hw,act,residual,two_place", FUSED_CASES)
def test_conv_bn_act_node_matches_the_separate_modules(cin, cout, hw, act, residual, two_place):
    """ops._PwConvBNActFn (one autograd node: convolution + BatchNorm + activation, its backward on hd_bn_act_bwd_* and
    hd_pointwise_bwd) through layers.ConvBNAct under autocast against the same layer with the 1x1 kernels switched off
    (library convolution + fused BatchNorm): ragged rows, post-activation residual, output in two places (the merged C3
    branch convolutions), a Cin > 256 layer, 32-channel tiles.  Two runs of float atomics and two convolution
    implementations: equal up to a couple of bf16 roundings of the element or of a typical one — except under ReLU, whose
    derivative flips for pre-activations that the two paths round to different sides of zero (single elements then differ
    by their whole value): those cases are bounded in relative L2 instead.  The bound: the two convolutions' outputs differ by
    up to one bf16 rounding (2^-8 of |y|), so a normalised pre-activation z ~ N(0, 1) changes sign with probability about
    2 pdf(0) 2^-8 E|z| = 2.5e-3; every flipped element loses or gains its whole gradient, i.e. a relative L2 error of about
    sqrt(2.5e-3) = 5e-2 in dz and in everything computed from it.  (First written as 2e-2 without this estimate; the first
    GPU run measured 2.6e-2 on the 32 -> 96 case and the number above is the estimate, not that measurement.)"""
    from heltondetection_amd import layers
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = layers.ConvBNAct(cin, cout, 1, 1, act=act).to(dev).to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        m.bn.weight.uniform_(0.5, 1.5); m.bn.bias.uniform_(-0.5, 0.5)
    B = 3
    x0 = torch.randn((B, cin, hw, hw), device=dev).contiguous(memory_format=torch.channels_last)
    res_in = torch.randn((B, cout, hw, hw), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if residual else None
    if two_place:
        c1 = cout // 2
        gy = (torch.randn((B, c1, hw, hw), device=dev), torch.randn((B, cout - c1, hw, hw), device=dev))
    else:
        gy = torch.randn((B, cout, hw, hw), device=dev).contiguous(memory_format=torch.channels_last)
    runs = {}
    was = layers.ConvBNAct.pointwise
    for ours in (True, False):
        layers.ConvBNAct.pointwise = ours
        try:
            out_pair = None
            if two_place:   # first half dense, second half a slice of a wider concatenation buffer
                y1 = torch.empty((B, c1, hw, hw), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
                cat = torch.zeros((B, 2 * cout, hw, hw), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
                out_pair = (y1, cat[:, cout:cout + (cout - c1)])
            runs[ours] = _layer_run(m, x0, gy, res_in, out_pair, dev)
        finally:
            layers.ConvBNAct.pointwise = was
    for k in runs[True]:
        a, b = runs[True][k], runs[False][k]
        if a is None:
            continue
        if act == "relu" and k in ("dx", "dw", "dgamma", "dbeta", "dres"):
            rel = float((a - b).norm() / (b.norm() + 1e-12))
            assert rel <= 5e-2, (k, rel)
            continue
        tol = 2.0 ** -5 * b.abs() + 2.0 ** -5 * b.abs().mean() + 1e-7
        assert bool(((a - b).abs() <= tol).all()), (k, float(((a - b).abs() / tol).max()))


@pytest.mark.parametrize("cin,cout,hw", [(64, 64, 20), (128, 256, 10), (256, 128, 8)])
def test_fused_conv_bn_act_matches_fp32_pytorch(cin, cout, hw):
    """The fused node against plain float32 PyTorch conv -> BatchNorm2d -> SiLU on the same bf16-rounded x and w: forward,
    dx, dw, dgamma, dbeta and the running statistics, element-wise with the bf16 tolerance of the kernels above."""
    from heltondetection_amd import layers, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = layers.ConvBNAct(cin, cout, 1, 1).to(dev).to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        m.conv.weight.copy_(m.conv.weight.to(torch.bfloat16).float())   # the reference sees the same weights
        m.bn.weight.uniform_(0.5, 1.5); m.bn.bias.uniform_(-0.5, 0.5)
    x0 = torch.randn((4, cin, hw, hw), device=dev).to(torch.bfloat16).float().contiguous(memory_format=torch.channels_last)
    gy = torch.randn((4, cout, hw, hw), device=dev).to(torch.bfloat16).float().contiguous(memory_format=torch.channels_last)
    got = _layer_run(m, x0, gy, None, None, dev)
    ref_conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev)
    ref_bn = torch.nn.BatchNorm2d(cout, eps=m.bn.eps, momentum=m.bn.momentum).to(dev).train()
    with torch.no_grad():
        ref_conv.weight.copy_(m.conv.weight); ref_bn.weight.copy_(m.bn.weight); ref_bn.bias.copy_(m.bn.bias)
    xr = x0.clone().requires_grad_(True)
    yr = torch.nn.functional.silu(ref_bn(ref_conv(xr)))
    (yr * gy).sum().backward()
    want = {"y": yr.detach(), "dx": xr.grad, "dw": ref_conv.weight.grad, "dgamma": ref_bn.weight.grad, "dbeta": ref_bn.bias.grad,
            "running_mean": ref_bn.running_mean, "running_var": ref_bn.running_var}
    # bf16 stores on the way: conv output (2^-9), activation output (2^-9), dy (2^-9) and the gradient itself; the
    # element-wise bound is a few bf16 roundings of the element plus a floor for elements that are sums of many terms
    for k, ref in want.items():
        a = got[k]
        tol = 2.0 ** -6 * ref.abs() + 2.0 ** -6 * ref.abs().mean() + 1e-6
        bad = float(((a - ref).abs() / tol).max())
        assert bad <= 1.0, (k, bad, float((a - ref).abs().max()), float(ref.abs().max()))


def test_chained_weight_gradient_sums_ride_in_the_next_launch():
    """Inside a training step (ops._WgradChain.enabled) the sum over the row slices of weight gradient i rides in the
    launch of weight gradient i + 1 and the last one is flushed: the results equal the immediately summed ones bit for bit,
    across different shapes in one chain and with a 3x3 weight gradient (its own two launches) in between."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    shapes = [(2, 64, 20, 20, 64), (2, 128, 10, 10, 256), (1, 256, 9, 7, 128), (2, 64, 40, 40, 128), (2, 512, 5, 5, 256)]
    ops_in = []
    for B, K, H, W, N in shapes:
        x = torch.randn((B, K, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn((B, N, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ops_in.append((dy, x))
    x3 = torch.randn((2, 64, 20, 20), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy3 = torch.randn((2, 64, 20, 20), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref = [ops.pointwise_wgrad_rows(dy, x).clone() for dy, x in ops_in]
    ref3 = ops.conv3x3_wgrad_rows(dy3, x3).clone()
    ops._WgradChain.enabled = True
    try:
        got = []
        for i, (dy, x) in enumerate(ops_in):
            got.append(ops.pointwise_wgrad_rows(dy, x))
            if i == 1:
                got3 = ops.conv3x3_wgrad_rows(dy3, x3)
        ops.flush_wgrad()
    finally:
        ops._WgradChain.enabled = False
    torch.cuda.synchronize()
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert torch.equal(got3, ref3)


@pytest.mark.parametrize("shape", [(2, 64, 20, 20, 64), (3, 128, 13, 11, 128), (2, 256, 10, 10, 64), (1, 64, 40, 40, 256),
                                   (2, 1024, 13, 13, 256), (1, 768, 9, 7, 256)])
def test_pointwise_dgrad_with_the_shortcut_gradient_added_in_the_epilogue(shape):
    """hd_pointwise_conv_dgrad_add == hd_pointwise_conv_dgrad followed by a bf16 add (float32 sum of the two bf16 values, one
    rounding), bit for bit; also with the result written over the addend."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    g = torch.Generator().manual_seed(K * N + H)
    dy = torch.randn((B, N, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((N, K), generator=g) * N ** -0.5).to(dev).to(torch.bfloat16)
    r = torch.randn((B, K, H, W), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    plain = ops.pointwise_dgrad_rows(dy, w)
    want = (plain.float() + r.float()).to(torch.bfloat16)
    got = ops.pointwise_dgrad_rows(dy, w, add=r)
    assert torch.equal(got, want)
    r2 = r.clone()
    ops.pointwise_dgrad_rows(dy, w, out=r2, add=r2)
    assert torch.equal(r2, want)


@pytest.mark.parametrize("c,hw", [(64, 20), (128, 13)])
def test_bottleneck_shortcut_gradient_through_the_fork_matches_autograd_add(c, hw):
    """layers.Bottleneck with the shortcut: the gradient of the block input is cv1's data gradient + the shortcut's gradient,
    summed inside the data-gradient kernel (fork) or by the autograd engine — same values up to the float atomics of the
    BatchNorm sums."""
    from heltondetection_amd import layers, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = layers.Bottleneck(c, c, True, e=1.0).to(dev).to(memory_format=torch.channels_last).train()
    x0 = torch.randn((3, c, hw, hw), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((3, c, hw, hw), device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        layers.ConvBNAct.fork_shortcut = on
        try:
            m.zero_grad(set_to_none=True)
            ops.bn_arena_reset(dev)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * gy).sum().backward()
            ops.flush_wgrad()
            res[on] = (y.detach().float(), x.grad.float(), m.cv1.conv.weight.grad.float().clone(), m.cv2.conv.weight.grad.float().clone())
        finally:
            layers.ConvBNAct.fork_shortcut = True
    # two runs: the BatchNorm sums' float atomics differ in their last bits, which moves single bf16 roundings of the
    # operands; dx is a SUM of two gradients that partly cancel, so the floor is a rounding of a typical operand
    for a, b in zip(res[True], res[False]):
        tol = 2.0 ** -6 * b.abs() + 2.0 ** -6 * b.abs().mean() + 1e-7
        assert bool(((a - b).abs() <= tol).all()), float(((a - b).abs() / tol).max())


# ---- the DEEP (long-launch) form of the streaming kernels: >= 6 row blocks per wave, K >= 128 ------------------------------
# (B, Cin, H, W, Cout): 128-wide column tiles for K = 256 (rows >= 40 000), the 64-wide ones below that, K = 128 at 208 x 208,
# ragged row counts (the peeled last blocks), many column tiles
DEEP_SHAPES = [(16, 256, 104, 104, 256), (1, 256, 419, 413, 256), (3, 256, 113, 117, 512), (16, 128, 208, 208, 128),
               (1, 128, 833, 829, 128), (2, 256, 157, 127, 64)]


@pytest.mark.parametrize("shape", DEEP_SHAPES)
def test_deep_streaming_form_is_exact_on_integer_data_forward_gradient_and_fused_add(shape):
    """Small integers (exact sums in float32, results exactly representable in bf16): the long-launch form of
    hd_pointwise_conv / _dgrad / _dgrad_add — two row blocks of lookahead, peeled guarded blocks, fragments two k-steps ahead —
    must reproduce the float32 matmul exactly, statistics included, on full and ragged row counts."""
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    g = torch.Generator().manual_seed(K + N + H)
    x = torch.randint(-2, 3, (B, K, H, W), generator=g, dtype=torch.int8).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randint(-1, 2, (N, K), generator=g, dtype=torch.int8).to(dev).to(torch.bfloat16)
    # sparse weights keep most results below 256, where every integer is a bf16 value (larger ones round once, as the reference)
    keep = (torch.rand((N, K), generator=g) < 64.0 / K).to(dev)
    w = torch.where(keep, w, torch.zeros_like(w))
    lib = _lib.load()
    rep = lib.hd_bn_act_replicas()
    stats = torch.zeros(rep * 2 * N, dtype=torch.float32, device=dev)
    y = ops.pointwise_conv_rows(x, w, stats)
    R = B * H * W
    xr = x.permute(0, 2, 3, 1).reshape(R, K).float()
    ref = (xr @ w.float().t()).to(torch.bfloat16).float()  # exact integer sums, one rounding to bf16 (exact up to 256)
    got = y.permute(0, 2, 3, 1).reshape(R, N).float()
    assert torch.equal(got, ref)
    want = torch.stack([ref.double().sum(0), (ref.double() ** 2).sum(0)])
    have = stats.view(rep, 2, N).double().sum(0)
    assert float(((have - want).abs() / (want.abs() + 1.0)).max()) < 1e-6
    # data gradient of the same layer (dy has N channels, the result K) and its fused-add form
    if lib.hd_pointwise_conv_dgrad_supported(R, K, N):
        dy = torch.randint(-2, 3, (B, N, H, W), generator=g, dtype=torch.int8).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dref = (dy.permute(0, 2, 3, 1).reshape(R, N).float() @ w.float()).to(torch.bfloat16).float()
        dx = ops.pointwise_dgrad_rows(dy, w)
        assert torch.equal(dx.permute(0, 2, 3, 1).reshape(R, K).float(), dref)
        r = torch.randint(-3, 4, (B, K, H, W), generator=g, dtype=torch.int8).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dxa = ops.pointwise_dgrad_rows(dy, w, add=r)
        want_a = (dref + r.permute(0, 2, 3, 1).reshape(R, K).float()).to(torch.bfloat16).float()
        assert torch.equal(dxa.permute(0, 2, 3, 1).reshape(R, K).float(), want_a)


# ---- data gradients of the Cout = 512 layers: hd_pointwise_conv on the transposed weight (round 5) -----------------------------------
@pytest.mark.parametrize("cin,cout,hw", [(512, 512, 20), (256, 512, 13), (1024, 512, 20)])
def test_long_output_layer_data_gradient_runs_on_the_transposed_weight(cin, cout, hw):
    """Cout = 512 is too long for hd_pointwise_conv_dgrad's transpose fragments: with the layer's weight registered in
    ops.FlippedWeights (taps = 1: one transposing launch per step for all such layers) the data gradient is hd_pointwise_conv(dy, w^T)
    on the long-reduction kernel; without the registration the library computes it.  Both must agree with the float32 reference."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B = 16 if hw == 20 else 3
    g = torch.Generator().manual_seed(cin + cout + hw)
    x = torch.randn((B, cin, hw, hw), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 1, 1), generator=g) * cin ** -0.5).to(dev).to(torch.bfloat16)
    gy = torch.randn((B, cout, hw, hw), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref = torch.nn.functional.conv2d(gy.float(), w.float().permute(1, 0, 2, 3))   # dx = dy . w over Cout
    fl = ops.FlippedWeights([w])
    assert len(fl) == 1
    fl.refresh()
    assert torch.equal(fl.by_ptr[w.data_ptr()].view(cin, cout), w.view(cout, cin).t())   # the flip launch with taps = 1: a transpose
    got = {}
    for registered in (True, False):
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        ops.enable_timing("pointwise_dgrad")
        if registered:
            fl2 = ops.FlippedWeights([wr.detach()])   # the registration is by storage: the trainer registers its bf16 shadows
            fl2.by_ptr[wr.data_ptr()] = fl2.by_ptr.pop(wr.detach().data_ptr()) if wr.detach().data_ptr() != wr.data_ptr() else fl2.by_ptr[wr.data_ptr()]
            fl2.refresh()
            with fl2:
                y = ops.pointwise_conv(xr, wr, None)
                y.backward(gy)
        else:
            y = ops.pointwise_conv(xr, wr, None)
            y.backward(gy)
        ops.flush_wgrad()
        torch.cuda.synchronize()
        ran = len(ops.collect_timing()["pointwise_dgrad"])
        ops.disable_timing()
        assert ran == (1 if registered else 0)
        got[registered] = xr.grad.float()
    tol = 2.0 ** -8 * ref.abs() + 2e-5 * ref.abs().max()
    assert bool(((got[True] - ref).abs() <= tol * 1.01).all()), float((got[True] - ref).abs().max())
    assert float((got[False] - ref).abs().max() / ref.abs().max()) < 2e-2   # the library's path (bf16 output, its own order)
