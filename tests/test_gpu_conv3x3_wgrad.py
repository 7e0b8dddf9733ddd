"""hd_conv3x3_wgrad (csrc/pointwise_conv.hip): the 3x3 / stride 1 / pad 1 weight gradient through LDS transpose
fragments, nine taps as views of a three-row ring with a zero halo."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "flat"])
def wgrad_kernel(request):
    """The weight gradient has two kernels behind one entry point (row kernels / flat-position kernel, csrc/conv3x3_wgrad.hip);
    `flat` forces the second on every shape it can run, `auto` is the shipped choice."""
    from heltondetection_amd import _lib
    lib = _lib.load()
    lib.hd_conv3x3_wgrad_select(2 if request.param == "flat" else 0)
    try:
        yield request.param
    finally:
        lib.hd_conv3x3_wgrad_select(0)


def _ref(dy, x, stride=1):
    w = torch.zeros((dy.shape[1], x.shape[1], 3, 3), dtype=torch.float32, device=x.device)
    return torch.ops.aten.convolution_backward(dy.float(), x.float(), w, None, (stride, stride), (1, 1), (1, 1), False, (0, 0), 1,
                                               (False, True, False))[1]


@pytest.mark.parametrize("shape", [(2, 64, 20, 20, 64), (1, 128, 7, 20, 64), (3, 64, 40, 40, 128), (2, 128, 13, 40, 128),
                                   (1, 64, 80, 80, 64), (2, 64, 5, 80, 64), (16, 256, 20, 20, 256), (1, 64, 1, 20, 64),
                                   (2, 64, 26, 26, 128), (1, 128, 11, 52, 64), (1, 64, 9, 104, 64),
                                   # round 5 (flat position blocks): channel counts in multiples of 32, any width, many slices
                                   (2, 32, 9, 40, 32), (1, 96, 6, 12, 64), (2, 32, 5, 160, 32), (1, 32, 3, 208, 64), (3, 160, 17, 33, 96),
                                   (16, 128, 40, 40, 128), (4, 32, 160, 160, 32)])
def test_conv3x3_wgrad_is_exact_on_integer_data(shape, wgrad_kernel):
    """Small integers: all products and sums are exact in float32, so borders (zero halo), row slices with their halo
    rows, the pixel padding to 16 and the tap -> (row slot, shift) mapping must reproduce the exact gradient."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    g = torch.Generator().manual_seed(K + N + H + W)
    x = torch.randint(-2, 3, (B, K, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randint(-2, 3, (B, N, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if not ops.conv3x3_wgrad_supported(x, N, 1):
        assert wgrad_kernel == "auto"      # a shape the shipped choice leaves to the library
        pytest.skip("left to the library by the automatic choice")
    dw = ops.conv3x3_wgrad_rows(dy, x)
    ref = _ref(dy, x)
    assert dw.shape == (N, K, 3, 3) and dw.dtype == torch.bfloat16
    assert torch.equal(dw.float(), ref.to(torch.bfloat16).float())


def test_conv3x3_layer_matches_library_path():
    from heltondetection_amd import layers, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = layers.ConvBNAct(64, 64, 3, 1).to(dev).to(memory_format=torch.channels_last).train()
    x0 = torch.randn((4, 64, 40, 40), device=dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((4, 64, 40, 40), device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        layers.ConvBNAct.pointwise = on
        layers.ConvBNAct.conv3x3_forward = on   # also the hand-written forward / data gradient
        try:
            m.zero_grad(set_to_none=True)
            ops.bn_arena_reset(dev)
            x = x0.clone().requires_grad_(True)
            ops.enable_timing("conv3x3_wgrad", "conv3x3_fwd")
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x.to(torch.bfloat16))
            (y.float() * gy).sum().backward()
            torch.cuda.synchronize()
            t_ = ops.collect_timing()
            assert len(t_["conv3x3_wgrad"]) == (1 if on else 0) and len(t_["conv3x3_fwd"]) == (1 if on else 0)
            ops.disable_timing()
            res[on] = (y.float().detach(), x.grad.float(), m.conv.weight.grad.float())
        finally:
            layers.ConvBNAct.pointwise = True
            layers.ConvBNAct.conv3x3_forward = True   # the default (restored as False until round 5: every later test of a suite
                                                      # run then took the library's 3x3 forward; tests/conftest.py now guards this)
    for name, a, b in zip(("y", "dx", "dw"), res[True], res[False]):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-6))
        assert err < 3e-2, (name, err)


@pytest.mark.parametrize("shape", [(2, 64, 20, 20, 64), (1, 128, 7, 20, 64), (2, 64, 40, 40, 128), (1, 64, 9, 40, 64),
                                   (1, 64, 80, 80, 128), (16, 256, 20, 20, 512), (1, 64, 1, 20, 64),
                                   (2, 64, 13, 26, 64), (1, 64, 7, 52, 128), (1, 64, 5, 104, 64),
                                   (2, 32, 9, 40, 64), (1, 96, 5, 11, 32), (2, 32, 6, 160, 64), (16, 128, 40, 40, 256), (3, 64, 7, 30, 96)])
def test_conv3x3_stride2_wgrad_is_exact_on_integer_data(shape, wgrad_kernel):
    """Stride 2: (B, Cin, Ho, Wo, Cout) with the input 2 Ho x 2 Wo; even / odd pixel planes, two new input rows per output row."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    g = torch.Generator().manual_seed(K + N + H + W + 1)
    x = torch.randint(-2, 3, (B, K, 2 * H, 2 * W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randint(-2, 3, (B, N, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if not ops.conv3x3_wgrad_supported(x, N, 2):
        assert wgrad_kernel == "auto"
        pytest.skip("left to the library by the automatic choice")
    dw = ops.conv3x3_wgrad_rows(dy, x, 2)
    ref = _ref(dy, x, 2)
    assert torch.equal(dw.float(), ref.to(torch.bfloat16).float())


@pytest.mark.parametrize("shape", [(2, 64, 20, 20, 64), (1, 128, 7, 20, 64), (2, 64, 40, 40, 128), (1, 128, 13, 40, 128),
                                   (1, 64, 80, 80, 64), (2, 64, 5, 80, 128), (16, 256, 20, 20, 256), (1, 64, 1, 20, 64),
                                   (3, 192, 9, 40, 64), (2, 128, 26, 26, 64), (1, 64, 11, 52, 128), (1, 64, 7, 104, 64),
                                   # csrc/conv3x3_c32.hip (32 -> 32 channels, 160-pixel rows): one row, two, a ragged walk, the bench shape
                                   (1, 32, 1, 160, 32), (1, 32, 2, 160, 32), (3, 32, 23, 160, 32), (16, 32, 160, 160, 32), (600, 32, 3, 160, 32)])
def test_conv3x3_forward_and_statistics_are_exact_on_integer_data(shape):
    """Small integers: exact sums, so the ring / halo / tap views, the Cin chunking, the pixel-tile padding and the
    transposed store must reproduce F.conv2d exactly; the epilogue's statistics are those of the tensor written."""
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    g = torch.Generator().manual_seed(K + N + H + W + 7)
    x = torch.randint(-2, 3, (B, K, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randint(-1, 2, (N, K, 3, 3), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    rep = _lib.load().hd_bn_act_replicas()
    stats = torch.zeros(rep * 2 * N, dtype=torch.float32, device=dev)
    y = ops.conv3x3_fwd_rows(x, w, stats)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, 1, 1)
    assert torch.equal(y.float(), ref.to(torch.bfloat16).float())
    yf = y.double()
    want = torch.stack([yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))])
    got = stats.view(rep, 2, N).double().sum(0)
    assert float(((got - want).abs() / (want.abs() + 1.0)).max()) < 1e-5
    # the data gradient through the same kernel with flipped, transposed weights
    dy = torch.randint(-2, 3, (B, N, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
    dx = ops.conv3x3_fwd_rows(dy, wt)
    dref = torch.ops.aten.convolution_backward(dy.float(), x.float(), w.float(), None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                               (True, False, False))[0]
    assert torch.equal(dx.float(), dref.to(torch.bfloat16).float())


@pytest.mark.parametrize("shape", [(2, 96, 9, 40, 64), (1, 32, 6, 20, 64), (2, 160, 4, 80, 128), (1, 32, 3, 104, 64), (2, 96, 5, 26, 64),
                                   (2, 32, 9, 160, 32)])
def test_conv3x3_forward_takes_input_channels_in_multiples_of_32_and_channel_slices(shape):
    """Round 5: the kernel walks Cin in 32-channel chunks, so Cin = 32, 96, 160 run (the round-4 kernel needed multiples of 64);
    x and y as channel slices of wider channels-last tensors (row stride > channels), the neighbours of y untouched."""
    from heltondetection_amd import _lib, ops
    import ctypes as C
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    g = torch.Generator().manual_seed(K + N + H + W)
    xw = torch.randint(-2, 3, (B, K + 32, H, W), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = xw[:, 16:16 + K] if K % 16 == 0 else xw[:, :K]
    w = torch.randint(-1, 2, (N, K, 3, 3), generator=g).float().to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ops.conv3x3_fwd_supported(x, N)
    y = ops.conv3x3_fwd_rows(x, w)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, 1, 1)
    assert torch.equal(y.float(), ref.to(torch.bfloat16).float())
    # output into a channel slice
    buf = torch.full((B, N + 64, H, W), 5.0, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    lib = _lib.load()
    out = buf[:, 32:32 + N]
    ops.check(lib.hd_conv3x3_fwd(ops._ptr(x), int(x.stride(3)), ops._ptr(w), B, H, W, K, N, ops._ptr(out), int(out.stride(3)), None,
                                 ops._stream(x)), "hd_conv3x3_fwd")
    torch.cuda.synchronize()
    assert torch.equal(out.float(), ref.to(torch.bfloat16).float())
    assert bool((buf[:, :32] == 5.0).all()) and bool((buf[:, 32 + N:] == 5.0).all())


def test_flipped_weights_of_all_layers_in_one_launch_equal_flip_permute():
    """hd_conv3x3_flip_weights (ops.FlippedWeights): wt[c][ky][kx][n] = w[n][2 - ky][2 - kx][c] for a batch of layers with
    different (ragged) channel counts, bit for bit what flip + permute + contiguous builds; consulted only inside `with`."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn((n, c, 3, 3), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
          for n, c in ((64, 64), (128, 64), (64, 256), (96, 40), (256, 256), (32, 33))]
    fl = ops.FlippedWeights(ws)
    assert len(fl) == len(ws)
    fl.refresh()
    assert ops.FlippedWeights.get(ws[0]) is None            # not active outside the window
    with fl:
        for w in ws:
            got = ops.FlippedWeights.get(w)
            want = w.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
            assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
            assert torch.equal(got, want)
    assert ops.FlippedWeights.get(ws[0]) is None
