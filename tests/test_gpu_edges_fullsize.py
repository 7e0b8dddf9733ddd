"""Edge cases through the C-ABI and size-independent properties at BASELINE.json's full sizes
(the serial oracle would need minutes there, so the checks are identities between HIP paths, or
against plain torch fp32 ops on the GPU for the floating-point decode)."""
import math

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from heltondetection_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------------------- edge cases
def test_empty_and_degenerate_inputs(gpu, ops):
    z4 = torch.zeros((2, 0, 4), device=gpu)
    b = H.boxes_batch(2, 5, 64, 1).to(gpu)
    assert ops.pairwise_iou(z4, b).shape == (2, 0, 5)
    assert ops.pairwise_iou(b, z4).shape == (2, 5, 0)
    mx, am, gmx, gam = ops.iou_assign(b, z4, None)
    assert (mx == -1).all() and (am == -1).all() and gmx.shape == (2, 0)
    mx, am, gmx, gam = ops.iou_assign(b, b, torch.zeros(2, dtype=torch.int32, device=gpu))   # all gt padded out
    assert (mx == -1).all() and (am == -1).all() and (gmx == -1).all() and (gam == -1).all()
    keep, cnt = ops.nms_batched(b, torch.rand((2, 5), device=gpu), None, torch.zeros(2, dtype=torch.int32, device=gpu), 0.0, 0.5, 0, 5)
    assert (cnt == 0).all() and (keep == -1).all()
    keep, cnt = ops.nms_batched(b, torch.full((2, 5), float("nan"), device=gpu), None, None, 0.0, 0.5, 0, 5)
    assert (cnt == 0).all()                                            # NaN scores never pass
    keep, cnt = ops.nms_batched(b, torch.rand((2, 5), device=gpu), None, None, 0.0, 0.5, 0, 0)
    assert keep.shape == (2, 0) and (cnt == 0).all()
    # zero-area boxes never suppress each other
    zb = torch.tensor([[[5., 5., 5., 5.], [5., 5., 5., 5.], [1., 1., 9., 9.]]], device=gpu)
    keep, cnt = ops.nms_batched(zb, torch.tensor([[0.9, 0.8, 0.7]], device=gpu), None, None, 0.0, 0.1, 0, 3)
    assert int(cnt[0]) == 3
    # RoIAlign with no RoIs, WBF with nothing above the skip threshold
    f = [torch.randn((1, 8, 16, 16), device=gpu)]
    assert ops.roi_align(f, torch.zeros((0, 5), device=gpu), 3, [1.0], 2, True, 2, 224.0, 4).shape == (0, 8, 3, 3)
    assert ops.roi_pool(f, torch.zeros((0, 5), device=gpu), 3, [1.0], 2, 224.0, 4).shape == (0, 8, 3, 3)
    out, c = ops.wbf(b, torch.zeros((2, 5), device=gpu), torch.zeros((2, 5), dtype=torch.int32, device=gpu),
                     torch.ones((2, 5), device=gpu), None, 1.0, 0.5, 0.1)
    assert (c == 0).all() and float(out.abs().sum()) == 0.0


def test_ragged_batches_match_oracle(gpu, ops):
    """Every image has a different number of valid gt boxes / candidates (padding must be inert)."""
    B, N, M = 4, 700, 48
    a = H.boxes_batch(B, N, 512, 3)
    b = H.boxes_batch(B, M, 512, 4)
    mv = torch.tensor([48, 1, 0, 17], dtype=torch.int32)
    got = ops.iou_assign(a.to(gpu), b.to(gpu), mv.to(gpu))
    ref = co.iou_assign(a.numpy(), b.numpy(), mv.numpy())
    for x, y in zip(got, ref):
        assert np.array_equal(x.cpu().numpy(), y)
    scores = H.distinct_scores(B, N, 5)
    nv = torch.tensor([700, 3, 0, 65], dtype=torch.int32)   # 65: one past a 64-box tile boundary
    keep, cnt = ops.nms_batched(a.to(gpu), scores.to(gpu), None, nv.to(gpu), 0.1, 0.4, 0, 128)
    rk, rc = co.nms_batched(a.numpy(), scores.numpy(), None, nv.numpy(), 0.1, 0.4, 0, 128)
    assert np.array_equal(cnt.cpu().numpy(), rc) and np.array_equal(keep.cpu().numpy(), rk)


def test_status_codes_on_bad_arguments(gpu, ops):
    import ctypes as C
    from heltondetection_amd import _lib
    lib = _lib.load()
    t = torch.zeros((1, 70000, 4), device=gpu)
    s = torch.zeros((1, 70000), device=gpu)
    with pytest.raises(_lib.HdError, match="out of range"):            # > 65536 candidates with no clamp
        ops.nms_batched(t, s, None, None, -1.0, 0.5, 0, 10)
    keep, cnt = ops.nms_batched(t, s, None, None, -1.0, 0.5, 4096, 10)  # clamped: fine
    assert keep.shape == (1, 10)
    with pytest.raises(_lib.HdError):                                   # dtype contract
        ops.pairwise_iou(t.double(), t)
    # workspace too small is reported, not written past
    ws = torch.empty(16, dtype=torch.uint8, device=gpu)
    k = torch.empty((1, 4), dtype=torch.int32, device=gpu)
    c = torch.empty((1,), dtype=torch.int32, device=gpu)
    st = lib.hd_nms_batched(C.c_void_p(t.data_ptr()), C.c_void_p(s.data_ptr()), None, None, 1, 100, 0.0, 0.5, 0, 4,
                            C.c_void_p(k.data_ptr()), C.c_void_p(c.data_ptr()), C.c_void_p(ws.data_ptr()), 16, None)
    assert st == -4


# ------------------------------------------------------------------------------- full-size properties
def test_fullsize_fused_assign_equals_materialised_reductions(gpu, ops):
    """BASELINE config 5 shape: 1280^2, A=3 -> N=409200 anchors, ~500 gt.  max/argmax of the materialised
    [B,N,M] matrix (torch reductions on the GPU) must equal the fused kernel's outputs bit for bit."""
    from heltondetection_amd import synthetic as syn
    g = syn.make_generator(21)
    B, N, M = 2, 409200, 500
    a = syn.random_boxes(N, 1280.0, g, 8.0, 512.0).to(gpu)
    b = torch.stack([syn.random_boxes(M, 1280.0, g, 4.0, 64.0) for _ in range(B)]).to(gpu)
    mv = torch.tensor([500, 433], dtype=torch.int32, device=gpu)
    iou = ops.pairwise_iou(a, b, mv)                   # 1.6 GB
    mx, am, gmx, gam = ops.iou_assign(a, b, mv)
    assert float(iou.max()) <= 1.0 and float(iou[0].min()) >= 0.0
    assert (iou[1, :, 433:] == -1).all()
    rmx, _ = iou.max(dim=2)
    assert torch.equal(rmx, mx)
    # argmax: the fused index must point at a maximal entry, and be the FIRST such entry
    picked = torch.gather(iou, 2, am.long().clamp(min=0)[..., None])[..., 0]
    assert torch.equal(picked, mx)
    first = (iou == mx[..., None]).int().argmax(dim=2)
    assert torch.equal(first.int(), am)
    cmx, _ = iou.max(dim=1)
    assert torch.equal(cmx[0], gmx[0]) and torch.equal(cmx[1, :433], gmx[1, :433]) and (gmx[1, 433:] == -1).all()
    cfirst = (iou == cmx[:, None, :]).int().argmax(dim=1)
    assert torch.equal(cfirst[0].int(), gam[0]) and torch.equal(cfirst[1, :433].int(), gam[1, :433])
    del iou


def test_fullsize_decode_matches_torch_fp32(gpu, ops):
    """bs16 640x640 C=80 through the channels-last view, against plain torch fp32 ops on the GPU (1e-4 rel)."""
    from heltondetection_amd.heads import YOLOV5_ANCHORS
    B, no = 16, 85
    g = torch.Generator(device="cpu").manual_seed(31)
    views, ref_rows = [], []
    for l, s in enumerate((8, 16, 32)):
        Hh = 640 // s
        conv = (torch.randn((B, 3 * no, Hh, Hh), generator=g) * 2).to(gpu).contiguous(memory_format=torch.channels_last)
        v = conv.view(B, 3, no, Hh, Hh).permute(0, 1, 3, 4, 2)
        views.append(v)
        y = v.sigmoid()
        yy, xx = torch.meshgrid(torch.arange(Hh, device=gpu, dtype=torch.float32), torch.arange(Hh, device=gpu, dtype=torch.float32), indexing="ij")
        grid = torch.stack((xx, yy), 2).view(1, 1, Hh, Hh, 2)
        anc = torch.tensor(YOLOV5_ANCHORS[l], device=gpu, dtype=torch.float32).view(1, 3, 1, 1, 2)
        ref_rows.append(torch.cat(((y[..., :2] * 2 - 0.5 + grid) * s, (y[..., 2:4] * 2) ** 2 * anc, y[..., 4:]), -1).reshape(B, -1, no))
    ref = torch.cat(ref_rows, 1)
    out = ops.yolo_decode(views, (8, 16, 32), YOLOV5_ANCHORS)
    assert out.shape == (16, 25200, 85)
    rel = ((out - ref).abs() / (ref.abs() + 1e-4)).max()
    assert float(rel) < 1e-4


def test_fullsize_loss_bookkeeping(gpu, ops):
    """bs16 640x640: total == B*(lbox+lobj+lcls); the dense gradient is zero outside channel 4 except at the
    matched cells; its per-channel sums equal the bias gradient the kernels report."""
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.heads import YOLOV5_ANCHORS
    B, no = 16, 85
    g = syn.make_generator(41)
    gtb, gtc, mv = syn.make_targets(B, 640, g, "coco", 80, 40)
    convs = []
    for s in (8, 16, 32):
        Hh = 640 // s
        convs.append(torch.randn((B, 3 * no, Hh, Hh), generator=g).to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True))
    views = [c.view(B, 3, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2) for c in convs]
    shapes = [(3, v.shape[2], v.shape[3], no) for v in views]
    slots = ops.yolo_assign(shapes, (8, 16, 32), YOLOV5_ANCHORS, gtb.to(gpu), gtc.to(gpu), mv.to(gpu))
    biases = [torch.zeros(3 * no, device=gpu, requires_grad=True) for _ in views]
    total, parts = ops.yolo_loss(views, (8, 16, 32), YOLOV5_ANCHORS, slots, 40, biases=biases)
    assert abs(float(parts[0]) - B * float(parts[1] + parts[2] + parts[3])) < 1e-3 * float(parts[0])
    assert abs(float(total) - float(parts[0])) < 1e-6
    total.backward()
    for l, c in enumerate(convs):
        gview = c.grad.view(B, 3, no, c.shape[2], c.shape[3])
        matched = torch.zeros((B * 3 * c.shape[2] * c.shape[3],), dtype=torch.bool, device=gpu)
        cells = slots[0][l]
        matched[cells[cells >= 0].long()] = True
        matched = matched.view(B, 3, c.shape[2], c.shape[3])
        others = gview[:, :, [0, 1, 2, 3] + list(range(5, no))]
        assert float(others.abs().amax(dim=2)[~matched].max()) == 0.0     # only matched cells carry box/class gradient
        ref_bias = gview.double().sum(dim=(0, 3, 4)).reshape(-1)
        assert float((biases[l].grad.double() - ref_bias).abs().max() / (ref_bias.abs().max() + 1e-12)) < 1e-4


def test_mosaic_edge_cases(gpu, ops):
    """No boxes at all, a one-image batch (all four tiles are the same image), pass-through for every image,
    an odd image size (rejected: the canvas is averaged 2x2), and the full 1280^2 size of config 5."""
    from heltondetection_amd import _lib, synthetic as syn
    from heltondetection_amd.augment import Mosaic
    g = syn.make_generator(171)
    S = 64
    img = syn.make_images(3, S, g).to(gpu)
    src = torch.tensor([[0, 1, 2, 0], [1, 1, 1, 1], [2, 0, 0, 1]], dtype=torch.int32, device=gpu)
    ctr = torch.full((3, 2), S, dtype=torch.int32, device=gpu)
    one = torch.ones(3, dtype=torch.int32, device=gpu)
    # M = 0
    out, ob, oc, cnt = ops.mosaic4(img, torch.zeros((3, 0, 4), device=gpu), torch.zeros((3, 0), dtype=torch.int32, device=gpu),
                                   torch.zeros(3, dtype=torch.int32, device=gpu), src, ctr, one)
    assert ob.shape == (3, 0, 4) and int(cnt.sum()) == 0 and out.shape == img.shape
    ref = co.mosaic4(img.cpu().numpy(), np.zeros((3, 1, 4), np.float32), np.zeros((3, 1), np.int32), np.zeros(3, np.int32),
                     src.cpu().numpy(), ctr.cpu().numpy(), one.cpu().numpy())
    assert np.array_equal(out.cpu().numpy(), ref[0])
    # B = 1: Mosaic.draw must not index outside the batch, and the op runs
    m = Mosaic(p=1.0, generator=torch.Generator(device=gpu).manual_seed(1))
    s1, c1, a1 = m.draw(1, S, gpu)
    assert s1.tolist() == [[0, 0, 0, 0]] and int(a1[0]) == 1
    gtb, gtc, mv = [t.to(gpu) for t in syn.make_targets(1, S, g, "coco", 5, 6)]
    o1 = ops.mosaic4(img[:1], gtb, gtc, mv, s1, c1, a1, min_wh=0.5)
    r1 = co.mosaic4(img[:1].cpu().numpy(), gtb.cpu().numpy(), gtc.cpu().numpy(), mv.cpu().numpy(), s1.cpu().numpy(),
                    c1.cpu().numpy(), a1.cpu().numpy(), min_wh=0.5)
    assert np.array_equal(o1[0].cpu().numpy(), r1[0]) and np.array_equal(o1[3].cpu().numpy(), r1[3])
    # apply == 0 everywhere: identity
    z = torch.zeros(3, dtype=torch.int32, device=gpu)
    own = torch.arange(3, dtype=torch.int32, device=gpu)[:, None].repeat(1, 4)
    outz = ops.mosaic4(img, torch.zeros((3, 2, 4), device=gpu), torch.zeros((3, 2), dtype=torch.int32, device=gpu),
                       torch.zeros(3, dtype=torch.int32, device=gpu), own, ctr, z)
    assert torch.equal(outz[0], img)
    # odd S is a shape error, not a silent wrong answer
    with pytest.raises(_lib.HdError):
        ops.mosaic4(torch.zeros((1, 3, 63, 63), device=gpu), gtb, gtc, mv, s1, c1, a1)
    # config-5 size: every output pixel of a p=1 mosaic is a mean of four [0,1) values or fill
    big = syn.make_images(2, 1280, g).to(gpu).contiguous(memory_format=torch.channels_last)
    sb, cb, ab = m.draw(2, 1280, gpu)
    ob_ = ops.mosaic4(big, torch.zeros((2, 1, 4), device=gpu), torch.zeros((2, 1), dtype=torch.int32, device=gpu),
                      torch.zeros(2, dtype=torch.int32, device=gpu), sb, cb, ab)[0]
    assert ob_.shape == big.shape and float(ob_.min()) >= 0.0 and float(ob_.max()) < 1.0
    assert abs(float(ob_.mean()) - 0.5) < 0.03      # mean of uniform noise (fill 0.447 covers little)


@pytest.mark.parametrize("seed", range(6))
def test_newest_ops_shape_sweep(gpu, ops, seed):
    """Seeded sweep over awkward shapes for the ops added last (RoI max-pooling, Mosaic-4, fused decode+NMS):
    every case must equal the oracle exactly (RoIPool, Mosaic) or the two-step HIP path exactly (detect)."""
    from heltondetection_amd import synthetic as syn
    rng = np.random.default_rng(1000 + seed)
    g = syn.make_generator(2000 + seed)
    # --- RoI max-pooling: odd channel counts, P in {1..7}, 1-3 levels, planar and channels-last features
    C = int(rng.choice([3, 8, 65, 130]))
    P = int(rng.integers(1, 8))
    L = int(rng.integers(1, 4))
    strides = [4, 8, 16][:L]
    B = int(rng.integers(1, 4))
    feats = [torch.randn((B, C, int(np.ceil(96 / s)), int(np.ceil(96 / s))), generator=g) for s in strides]
    R = int(rng.integers(1, 40))
    rb = syn.random_boxes(R, 96.0, g, 2.0, 90.0)
    rb[0] = torch.tensor([-5.0, -5.0, 120.0, 130.0])            # hangs off every side
    rois = torch.cat([torch.randint(0, B, (R, 1), generator=g).float(), rb], 1)
    fd = [f.to(gpu) for f in feats]
    if seed % 2:
        fd = [f.contiguous(memory_format=torch.channels_last) for f in fd]
    out, arg = ops.roi_pool(fd, rois.to(gpu), P, [1.0 / s for s in strides], 2, 56.0, 4, return_argmax=True)
    ref, rarg = co.roi_pool_fwd([f.numpy() for f in feats], rois.numpy(), P, [1.0 / s for s in strides], 2, 56.0, 4)
    assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(arg.cpu().numpy(), rarg)
    # --- Mosaic-4: S in {2, 6, 34, 50}, B 1..5, M 0..7, random centres over the whole allowed range
    S = int(rng.choice([2, 6, 34, 50]))
    Bm = int(rng.integers(1, 6))
    M = int(rng.integers(0, 8))
    img = syn.make_images(Bm, S, g)
    if M:
        gtb, gtc, mv = syn.make_targets(Bm, max(S, 16), g, "coco", 5, M)
        gtb = gtb.clamp(max=float(S))
    else:
        gtb, gtc, mv = torch.zeros((Bm, 0, 4)), torch.zeros((Bm, 0), dtype=torch.int32), torch.zeros(Bm, dtype=torch.int32)
    src = torch.from_numpy(rng.integers(0, Bm, (Bm, 4)).astype(np.int32))
    ctr = torch.from_numpy(rng.integers(S // 2, 3 * S // 2 + 1, (Bm, 2)).astype(np.int32))
    apl = torch.from_numpy(rng.integers(0, 2, (Bm,)).astype(np.int32))
    imd = img.to(gpu).contiguous(memory_format=torch.channels_last) if seed % 2 == 0 else img.to(gpu)
    o = ops.mosaic4(imd, gtb.to(gpu), gtc.to(gpu), mv.to(gpu), src.to(gpu), ctr.to(gpu), apl.to(gpu), min_wh=0.25)
    Mo = max(M, 1)
    r = co.mosaic4(img.numpy(), gtb.numpy() if M else np.zeros((Bm, 1, 4), np.float32),
                   gtc.numpy() if M else np.zeros((Bm, 1), np.int32), mv.numpy(), src.numpy(), ctr.numpy(), apl.numpy(),
                   min_wh=0.25)
    assert np.array_equal(o[0].cpu().numpy(), r[0]) and np.array_equal(o[3].cpu().numpy(), r[3])
    if M:
        assert np.array_equal(o[1].cpu().numpy(), r[1]) and np.array_equal(o[2].cpu().numpy(), r[2])
    # --- fused decode + NMS vs decode then NMS: class counts 1..90, odd image sizes, strided (channels-last) heads
    nc = int(rng.choice([1, 2, 20, 59, 90]))
    no = 5 + nc
    size = int(rng.choice([96, 128, 160]))
    Bd = int(rng.integers(1, 4))
    preds = H.yolo_preds(Bd, size, no, 3000 + seed, scale=2.5)
    views = []
    for p in preds:
        _, na, Hh, Ww, _ = p.shape
        if seed % 2:
            conv = p.permute(0, 1, 4, 2, 3).reshape(Bd, na * no, Hh, Ww).to(gpu).contiguous(memory_format=torch.channels_last)
            views.append(conv.view(Bd, na, no, Hh, Ww).permute(0, 1, 3, 4, 2))
        else:
            views.append(p.to(gpu))
    conf = float(rng.choice([0.01, 0.2, 0.5]))
    d1, c1 = ops.yolo_detect(views, H.YOLO_STRIDES, H.YOLO_ANCHORS, conf, 0.5, 4000, 64)
    d2, c2 = ops.yolo_nms(ops.yolo_decode(views, H.YOLO_STRIDES, H.YOLO_ANCHORS), conf, 0.5, 4000, 64)
    assert torch.equal(c1, c2) and torch.equal(d1, d2)


# ------------------------------------------------------------------------------- BASELINE config 3 at full size
def _cfg3_pyramid(gpu, dtype, fill=None):
    """P2..P5 of an 832x832 bs16 batch, 256 channels, channels-last (FasterRCNN-R50-PAFPN, config 3)."""
    g = torch.Generator(device=gpu).manual_seed(301)
    feats = []
    for s in (4, 8, 16, 32):
        shape = (16, 256, 832 // s, 832 // s)
        f = torch.randn(shape, device=gpu, generator=g) if fill is None else fill(shape, s)
        feats.append(f.to(dtype).contiguous(memory_format=torch.channels_last))
    return feats


def _cfg3_rois(gpu, R_per_img=512, inside=False):
    bx = H.boxes_batch(1, 16 * R_per_img, 832, 302, 16.0, 400.0)[0]
    if inside:  # keep every sample strictly inside the maps: 40 px margin (> one stride-32 cell)
        bx = bx.clamp(40.0, 792.0)
        bx[:, 2:] = torch.maximum(bx[:, 2:], bx[:, :2] + 8.0)
    bi = torch.arange(16).repeat_interleave(R_per_img)[:, None].float()
    return torch.cat([bi, bx], 1).to(gpu)


SCALES3 = [1 / 4, 1 / 8, 1 / 16, 1 / 32]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fullsize_roi_align_constant_and_ramp_identities(gpu, ops, dtype):
    """R = 8192, C = 256, P = 7, sampling 2, 832^2 bs16 channels-last.  Bilinear sampling reproduces constants and
    affine functions exactly (up to rounding) wherever no sample is clamped, so for RoIs inside the map:
    constant map c -> every bin == c;  f(y, x) = a*y + b*x + c0 on a level -> bin == f at the bin centre."""
    rois = _cfg3_rois(gpu, inside=True)
    const = _cfg3_pyramid(gpu, dtype, fill=lambda shape, s: torch.full(shape, 1.5, device=gpu))
    out = ops.roi_align(const, rois, 7, SCALES3, 2, True, 2, 224.0, 4, channels_last_out=True, out_dtype=torch.float32)
    assert out.shape == (8192, 256, 7, 7)
    assert float((out - 1.5).abs().max()) < 1e-5
    del const, out

    def ramp(shape, s):
        B, C, Hh, Ww = shape
        # a/16 and b/16 steps, exactly representable in bf16 over the map sizes used here (<= 208 cells: 13/16 steps)
        y = torch.arange(Hh, device=gpu, dtype=torch.float32)[:, None] * 0.125
        x = torch.arange(Ww, device=gpu, dtype=torch.float32)[None, :] * 0.0625
        return (y + x)[None, None].expand(B, C, Hh, Ww)

    feats = _cfg3_pyramid(gpu, dtype, fill=ramp)
    out = ops.roi_align(feats, rois, 7, SCALES3, 2, True, 2, 224.0, 4, channels_last_out=True, out_dtype=torch.float32)
    lvl = ops.roi_level_map(rois, 2, 4, 224.0, 4).long()
    sc = torch.tensor(SCALES3, device=gpu)[lvl]
    x1, y1, x2, y2 = [rois[:, i] * sc - 0.5 for i in (1, 2, 3, 4)]
    k = (torch.arange(7, device=gpu, dtype=torch.float32) + 0.5) / 7.0
    yc = y1[:, None] + (y2 - y1)[:, None] * k[None, :]        # bin centres in cell units
    xc = x1[:, None] + (x2 - x1)[:, None] * k[None, :]
    expect = yc[:, :, None] * 0.125 + xc[:, None, :] * 0.0625   # [R, 7, 7]
    err = (out - expect[:, None]).abs().max()
    # bf16 maps hold the ramp exactly only up to 8 significant bits: 26 * (2^-8) of headroom there
    assert float(err) < (1e-3 if dtype == torch.float32 else 0.15), float(err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fullsize_roi_align_adjoint(gpu, ops, dtype):
    """<fwd(x), g> == <x, bwd(g)> at config 3's full size (R = 8192 RoIs with borders and clamping included):
    the backward kernel is exactly the transpose of the forward one.  x and g are non-negative, so both sides are
    sums of non-negative terms and a relative tolerance is meaningful (with signed data the sums nearly cancel)."""
    feats = [f.requires_grad_(True) for f in
             _cfg3_pyramid(gpu, dtype, fill=lambda shape, s: torch.randn(shape, device=gpu, generator=torch.Generator(
                 device=gpu).manual_seed(300 + s)).abs())]
    rois = _cfg3_rois(gpu, inside=False)
    out = ops.roi_align(feats, rois, 7, SCALES3, 2, True, 2, 224.0, 4, channels_last_out=True, out_dtype=torch.float32)
    g = torch.randn(out.shape, device=gpu, generator=torch.Generator(device=gpu).manual_seed(303)).abs()
    g = g.contiguous(memory_format=torch.channels_last)
    lhs = float((out.detach().double() * g.double()).sum())
    out.backward(g)
    rhs = sum(float((f.detach().double() * f.grad.double()).sum()) for f in feats)
    # bf16: the returned gradients are rounded to bf16 once (2^-9 relative per element, random sign)
    assert lhs > 1e6
    assert abs(lhs - rhs) <= (2e-6 if dtype == torch.float32 else 1e-4) * lhs, (lhs, rhs)


def test_fullsize_rpn_proposals_equal_topk_decode_nms(gpu, ops):
    """hd_rpn_proposals at config 3's anchor set (832^2, 5 levels x 3 anchors, NA = 57 723 per image... x16 images,
    pre 2000 / post 1000) == per-level top-k -> hd_box_decode -> hd_nms_batched(cls = level) composed by hand;
    kept anchor indices compared exactly.  And NMS is idempotent on the result."""
    strides = [4, 8, 16, 32, 64]
    hw = [(math.ceil(832 / s), math.ceil(832 / s)) for s in strides]
    base = torch.stack([ops.anchor_base([8.0 * s], [0.5, 1.0, 2.0]) for s in strides])
    anchors = ops.anchor_grid(hw, strides, base, 0.5, gpu)
    ls = [h * w * 3 for h, w in hw]
    NA, B, pre, post = anchors.shape[0], 16, 2000, 1000
    g = torch.Generator(device=gpu).manual_seed(304)
    # distinct logits per image (ties would need the kernel's tie rule on the torch side too)
    logits = torch.stack([torch.randperm(NA, device=gpu, generator=g).float() for _ in range(B)]) / NA * 8.0 - 4.0
    deltas = torch.randn((B, NA, 4), device=gpu, generator=g) * 0.2
    rois, sc, cnt, idx = ops.rpn_proposals(logits, deltas, anchors, ls, pre, post, 0.7, 2.0, 832.0, 832.0, return_index=True)
    # by hand
    cand_idx, cand_lvl, off = [], [], 0
    for l, n in enumerate(ls):
        k = min(n, pre)
        top = torch.topk(logits[:, off:off + n], k, dim=1).indices + off
        cand_idx.append(top)
        cand_lvl.append(torch.full((B, k), l, dtype=torch.int32, device=gpu))
        off += n
    cand_idx, cand_lvl = torch.cat(cand_idx, 1), torch.cat(cand_lvl, 1)
    K = cand_idx.shape[1]
    a_sel = anchors[cand_idx.reshape(-1)]
    d_sel = torch.gather(deltas, 1, cand_idx[..., None].expand(-1, -1, 4)).reshape(-1, 4)
    boxes = ops.box_decode(a_sel.contiguous(), d_sel.contiguous(), 832.0, 832.0).view(B, K, 4)
    lg = torch.gather(logits, 1, cand_idx)
    ok = ((boxes[..., 2] - boxes[..., 0]) >= 2.0) & ((boxes[..., 3] - boxes[..., 1]) >= 2.0)
    score = torch.where(ok, lg, torch.full_like(lg, float("-inf")))
    keep, kc = ops.nms_batched(boxes.contiguous(), score.contiguous(), cand_lvl.contiguous(), None, float("-inf"), 0.7, 0, post)
    assert torch.equal(kc, cnt)
    kept_anchor = torch.where(keep >= 0, torch.gather(cand_idx, 1, keep.clamp(min=0).long()).to(torch.int32),
                              torch.full_like(keep, -1))
    assert torch.equal(kept_anchor, idx)
    sel = torch.gather(boxes, 1, keep.clamp(min=0).long()[..., None].expand(-1, -1, 4))
    sel = torch.where((keep >= 0)[..., None], sel, torch.zeros_like(sel))
    assert torch.equal(sel, rois)
    # idempotence: the survivors of one level never suppress each other
    lv_of = torch.gather(cand_lvl, 1, keep.clamp(min=0).long())
    sc_of = torch.gather(score, 1, keep.clamp(min=0).long())
    nv = cnt.clone()
    keep2, kc2 = ops.nms_batched(rois.contiguous(), sc_of.contiguous(), lv_of.contiguous(), nv, float("-inf"), 0.7, 0, post)
    assert torch.equal(kc2, cnt)
    ar = torch.arange(post, device=gpu, dtype=torch.int32)[None, :].expand(B, -1)
    assert torch.equal(keep2, torch.where(ar < cnt[:, None], ar, torch.full_like(ar, -1)))
