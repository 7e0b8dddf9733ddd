"""hd_upsample2x_cat (csrc/upcat.hip): nearest 2x upsampling + channel concatenation in one pass, forward and backward,
against torch's interpolate + cat (exact: the forward is a copy, the backward a sum of four values)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 16, 3, 5, 24), (1, 8, 1, 1, 8), (16, 256, 20, 20, 256), (3, 64, 7, 4, 128)])
def test_upsample2x_cat_forward_backward(dtype, shape):
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    B, Cs, Hs, Ws, Cb = shape
    g = torch.Generator().manual_seed(Cs + Cb + Hs)
    small = torch.randn((B, Cs, Hs, Ws), generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    big = torch.randn((B, Cb, 2 * Hs, 2 * Ws), generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((B, Cs + Cb, 2 * Hs, 2 * Ws), generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    s1, b1 = small.clone().requires_grad_(True), big.clone().requires_grad_(True)
    out = ops.upsample2x_cat(s1, b1)
    out.backward(gy)
    s2, b2 = small.clone().float().requires_grad_(True), big.clone().float().requires_grad_(True)
    ref = torch.cat((F.interpolate(s2, scale_factor=2.0, mode="nearest"), b2), 1)
    ref.backward(gy.float())
    assert out.dtype == dtype and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out.float(), ref.detach())
    assert torch.equal(b1.grad.float(), b2.grad)
    # four-term float32 sums, rounded once to the tensor dtype
    assert torch.equal(s1.grad, s2.grad.to(dtype)) or torch.allclose(s1.grad.float(), s2.grad, rtol=2.0 ** -7, atol=1e-6)


def test_upsample2x_cat_takes_channel_slices():
    from heltondetection_amd import ops
    dev = torch.device("cuda:0")
    wide = torch.randn((2, 64, 4, 4), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    big = torch.randn((2, 32, 8, 8), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    small = wide[:, 16:48]
    out = ops.upsample2x_cat(small, big)
    ref = torch.cat((F.interpolate(small.float(), scale_factor=2.0, mode="nearest"), big.float()), 1)
    assert torch.equal(out.float(), ref)
    assert not ops.upsample2x_cat_supported(small, big[:, :, :6])  # not 2x
    assert not ops.upsample2x_cat_supported(small.float(), big)    # mixed dtypes
