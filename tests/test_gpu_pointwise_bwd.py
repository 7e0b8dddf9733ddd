"""hd_pointwise_bwd (csrc/pointwise_bwd.hip): data gradient + weight-gradient partials of a 1x1 convolution from one staging
of dy.  Small-integer operands make every product and every partial sum exactly representable in float32, so the row
fragments and transpose fragments of the shared swizzled stage, the zeroed rows past the end, the fold of the four waves'
partials, the slab layout and the slice sum must reproduce the exact integer sums (then one bf16 rounding), bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ints(shape, lo, hi, g, dev, p_zero=0.0):
    t = torch.randint(lo, hi + 1, shape, generator=g).float()
    if p_zero > 0:
        t = t * (torch.rand(shape, generator=g) >= p_zero).float()
    return t.to(dev)


def _case(B, K, H, W, N, seed, dev):
    """dy [B, N, H, W], x [B, K, H, W] channels-last bf16 small integers; w [N, K] in {-1, 0, 1}."""
    g = torch.Generator().manual_seed(seed)
    dy = _ints((B, N, H, W), -2, 2, g, dev, 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = _ints((B, K, H, W), -3, 3, g, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = _ints((N, K), -1, 1, g, dev).to(torch.bfloat16)
    return dy, x, w


def _rows(t):  # [B, C, H, W] -> [R, C] float64
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).double()


def _check(dx, dw, dy, x, w, add=None):
    ref_dx = _rows(dy) @ w.double()
    if add is not None:
        # the kernel rounds dy.w to bf16 first, then adds the addend in float32 and rounds once more
        ref_dx = ref_dx.float().to(torch.bfloat16).double() + _rows(add)
    ref_dx = ref_dx.float().to(torch.bfloat16)
    got_dx = dx.permute(0, 2, 3, 1).reshape(-1, dx.shape[1])
    assert torch.equal(got_dx, ref_dx), float((got_dx.float() - ref_dx.float()).abs().max())
    ref_dw = (_rows(dy).t() @ _rows(x)).float().to(torch.bfloat16)
    assert torch.equal(dw.view_as(ref_dw), ref_dw), float((dw.view_as(ref_dw).float() - ref_dw.float()).abs().max())


# (B, Cin, H, W, Cout): every template instantiation (Cout 32 / 64 / 128 / 256 x column tiles of 32 / 64), ragged row counts
# (not multiples of 32 / 64), a single row, many column tiles (Cin 512 with Cout 256: 16), the headline model's layers
SHAPES = [(2, 32, 20, 20, 32), (1, 64, 17, 13, 32), (2, 64, 24, 20, 64), (1, 32, 9, 7, 64), (1, 96, 5, 5, 64),
          (2, 128, 9, 11, 128), (1, 64, 16, 16, 128), (1, 96, 7, 7, 128), (2, 256, 10, 10, 256), (1, 128, 7, 9, 256),
          (1, 512, 6, 6, 256), (1, 64, 1, 1, 64), (1, 256, 1, 3, 256), (16, 64, 80, 80, 64), (16, 128, 40, 40, 128),
          (16, 256, 20, 20, 256), (16, 512, 20, 20, 256), (4, 32, 160, 160, 32)]


@pytest.mark.parametrize("shape", SHAPES)
def test_pointwise_bwd_is_exact_on_integer_data(shape):
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    assert ops.pointwise_bwd_supported(B * H * W, K, N)
    dy, x, w = _case(B, K, H, W, N, 11 + K + N + H, dev)
    dx, dw = ops.pointwise_bwd_rows(dy, w, x)
    torch.cuda.synchronize()
    _check(dx, dw, dy, x, w)


@pytest.mark.parametrize("shape", [(2, 64, 12, 10, 64), (1, 128, 9, 9, 128), (2, 256, 7, 5, 256), (1, 32, 11, 3, 32)])
def test_pointwise_bwd_adds_the_shortcut_gradient_and_takes_channel_slices(shape):
    """add: the gradient arriving through a shortcut, summed in the epilogue (two bf16 values added in float32, one rounding).
    dy, x, add and dx as channel slices of wider channels-last tensors (row stride > channels)."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, K, H, W, N = shape
    g = torch.Generator().manual_seed(5 + K)
    dy, x, w = _case(B, K, H, W, N, 3 + N, dev)
    wide_dy = torch.zeros((B, N + 64, H, W), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    wide_dy[:, 32:32 + N] = dy
    wide_x = torch.zeros((B, K + 32, H, W), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    wide_x[:, :K] = x
    add = _ints((B, K, H, W), -4, 4, g, dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    buf = torch.full((B, K + 64, H, W), 7.0, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    out = buf[:, 32:32 + K]
    dx, dw = ops.pointwise_bwd_rows(wide_dy[:, 32:32 + N], w, wide_x[:, :K], add=add, out_dx=out)
    torch.cuda.synchronize()
    _check(dx, dw, dy, x, w, add)
    assert bool((buf[:, :32] == 7.0).all()) and bool((buf[:, 32 + K:] == 7.0).all())   # neighbours untouched


def test_pointwise_bwd_chains_slice_sums_like_the_weight_gradient_kernels():
    """Inside a chained backward pass (ops._WgradChain.enabled) the slice sum of layer L rides in extra workgroups of layer
    L + 1's launch — fused, plain 1x1 and 3x3 weight gradients in one chain — and the last one is flushed."""
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    cases = [_case(2, 64, 12, 12, 64, 1, dev), _case(1, 128, 10, 10, 256, 2, dev), _case(2, 256, 6, 6, 128, 3, dev)]
    leaves = [torch.zeros((c[2].shape[0], c[2].shape[1]), dtype=torch.bfloat16, device=dev, requires_grad=True) for c in cases]
    ops._WgradChain.enabled = True
    try:
        outs = [ops.pointwise_bwd_rows(dy, w, x, leaf=leaf) for (dy, x, w), leaf in zip(cases, leaves)]
        # a plain weight gradient in the same chain carries the last fused job
        dy, x, w = cases[0]
        dw_plain = ops.pointwise_wgrad_rows(dy, x, leaf=leaves[0])
        ops.flush_wgrad()
    finally:
        ops._WgradChain.enabled = False
        ops._WgradChain.reset()
    torch.cuda.synchronize()
    for (dy, x, w), (dx, dw) in zip(cases, outs):
        _check(dx, dw, dy, x, w)
    ref = (_rows(cases[0][0]).t() @ _rows(cases[0][1])).float().to(torch.bfloat16)
    assert torch.equal(dw_plain, ref)


def test_pointwise_bwd_rejects_what_it_cannot_take():
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda:0")
    assert not ops.pointwise_bwd_supported(100, 64, 512)      # Cout is the data gradient's reduction: 32 / 64 / 128 / 256
    assert not ops.pointwise_bwd_supported(100, 48, 64)       # Cin a multiple of 32
    dy, x, w = _case(1, 64, 4, 4, 64, 0, dev)
    with pytest.raises(_lib.HdError):
        ops.pointwise_bwd_rows(dy.float(), w, x)              # bf16 only
    with pytest.raises(_lib.HdError):
        ops.pointwise_bwd_rows(dy, w[:, :32].contiguous(), x)  # weight shape must match


@pytest.mark.parametrize("cin,cout,hw,fork", [(64, 64, 24, False), (128, 128, 12, True), (256, 256, 10, False), (512, 256, 6, False)])
def test_conv_bn_act_fused_backward_equals_the_two_kernel_backward(cin, cout, hw, fork):
    """The layer-level switch: ConvBNAct forward + backward with the one-launch backward on vs the dgrad + wgrad pair.  Same
    arithmetic (float32 accumulation of bf16 products, one rounding), different summation order over rows for dw only."""
    from heltondetection_amd import layers, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = layers.ConvBNAct(cin, cout, 1, 1).to(dev).to(memory_format=torch.channels_last).train()
    x0 = torch.randn((4, cin, hw, hw), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((4, cout, hw, hw), device=dev).contiguous(memory_format=torch.channels_last)
    gs = torch.randn((4, cin, hw, hw), device=dev).contiguous(memory_format=torch.channels_last)
    res = {}
    for on in (True, False):
        ops.PwBwdSwitch.enabled = on
        try:
            m.zero_grad(set_to_none=True)
            ops.bn_arena_reset(dev)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if fork:
                    y, xs = m(x, None, None, True)
                    loss = (y.float() * gy).sum() + (xs.float() * gs).sum()
                else:
                    loss = (m(x).float() * gy).sum()
            loss.backward()
            res[on] = (x.grad.float().clone(), m.conv.weight.grad.float().clone())
        finally:
            ops.PwBwdSwitch.enabled = True
    dx_on, dw_on = res[True]
    dx_off, dw_off = res[False]
    # the two runs do not see bit-identical dy: the BatchNorm backward's per-channel sums are float atomics whose order varies
    # (the kernels themselves are compared bit for bit on identical integer inputs above)
    for name, a, b in (("dx", dx_on, dx_off), ("dw", dw_on, dw_off)):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-9))
        assert err < 2.0 ** -7, (name, err)                  # one bf16 rounding
