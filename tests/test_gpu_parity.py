"""GPU parity: HIP kernels (through the C-ABI, heltondetection_amd.ops) vs the CPU oracle.

Bar (BASELINE.json north_star): bit-exact for anchor indices / argmax / NMS keep masks / slot
indices; 1e-4 relative for float outputs that go through expf/atanf/logf (device libm differs
from glibc by ulps).  The oracle is parity-unpinned (no reference source exists):
these tests prove "HIP path == our CPU restatement", not "== reference".
"""
import math

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def ops():
    from heltondetection_amd import ops as _ops
    return _ops


def dev(t, gpu):
    return t.to(gpu)


# ------------------------------------------------------------------------------- anchors
@pytest.mark.parametrize("size,A", [(640, 3), (832, 9)])
def test_anchor_grid_bit_exact(gpu, ops, size, A):
    strides = [4, 8, 16, 32, 64]
    hw = [(math.ceil(size / s), math.ceil(size / s)) for s in strides]
    ratios = [0.5, 1.0, 2.0]
    sizes = [[32.0 * 2 ** l] if A == 3 else [32.0 * 2 ** l * f for f in (1.0, 2 ** (1 / 3), 2 ** (2 / 3))] for l in range(5)]
    base = torch.stack([ops.anchor_base(sizes[l], ratios) for l in range(5)])
    ref_base = np.stack([co.anchor_base(sizes[l], ratios) for l in range(5)])
    assert np.array_equal(base.numpy(), ref_base)
    out = ops.anchor_grid(hw, strides, base, 0.5, gpu).cpu().numpy()
    ref = co.anchor_grid([h for h, _ in hw], [w for _, w in hw], strides, ref_base, 0.5)
    assert out.shape == ref.shape
    assert np.array_equal(out, ref)


# ------------------------------------------------------------------------------- IoU
@pytest.mark.parametrize("B,N,M", [(2, 1000, 20), (3, 777, 33), (1, 64, 1), (2, 5000, 500)])
def test_pairwise_iou_bit_exact(gpu, ops, B, N, M):
    a = H.boxes_batch(B, N, 640, 1)
    b = H.boxes_batch(B, M, 640, 2)
    mv = torch.tensor([M, max(M // 2, 0), M][:B], dtype=torch.int32)
    out = ops.pairwise_iou(dev(a, gpu), dev(b, gpu), dev(mv, gpu)).cpu().numpy()
    ref = co.pairwise_iou(a.numpy(), b.numpy(), mv.numpy())
    assert np.array_equal(out, ref)


def test_pairwise_iou_shared_anchors_and_large_m(gpu, ops):
    a = H.boxes_batch(1, 900, 1280, 3)[0]
    b = H.boxes_batch(2, 3500, 1280, 4, 4.0, 64.0)  # M > LDS capacity -> global path
    out = ops.pairwise_iou(dev(a, gpu), dev(b, gpu), None).cpu().numpy()
    ref = co.pairwise_iou(np.stack([a.numpy()] * 2), b.numpy(), None)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("B,N,M", [(2, 3000, 40), (2, 20000, 600), (1, 100, 0), (2, 5000, 2500)])
def test_iou_assign_bit_exact(gpu, ops, B, N, M):
    a = H.boxes_batch(B, N, 1280, 5, 4.0, 128.0)
    b = H.boxes_batch(B, max(M, 1), 1280, 6, 4.0, 64.0)[:, :M]
    mv = torch.tensor([M, M * 2 // 3][:B], dtype=torch.int32)
    mx, am, gmx, gam = ops.iou_assign(dev(a, gpu), dev(b.contiguous(), gpu), dev(mv, gpu))
    r_mx, r_am, r_gmx, r_gam = co.iou_assign(a.numpy(), b.numpy().reshape(B, M, 4), mv.numpy())
    assert np.array_equal(mx.cpu().numpy(), r_mx)
    assert np.array_equal(am.cpu().numpy(), r_am)
    assert np.array_equal(gmx.cpu().numpy(), r_gmx)
    assert np.array_equal(gam.cpu().numpy(), r_gam)


def test_iou_assign_all_zero_overlap(gpu, ops):
    # no anchor overlaps any gt: argmax must be 0 (first), gt_argmax 0, max 0
    a = torch.tensor([[[0., 0., 10., 10.], [20., 20., 30., 30.]]])
    b = torch.tensor([[[100., 100., 120., 120.], [200., 200., 220., 220.]]])
    mx, am, gmx, gam = ops.iou_assign(dev(a, gpu), dev(b, gpu), None)
    r = co.iou_assign(a.numpy(), b.numpy(), None)
    for x, y in zip((mx, am, gmx, gam), r):
        assert np.array_equal(x.cpu().numpy(), y)


# ------------------------------------------------------------------------------- codec / CIoU
def test_box_codec(gpu, ops):
    a = H.boxes_batch(1, 4096, 832, 7)[0]
    b = H.boxes_batch(1, 4096, 832, 8)[0]
    enc = ops.box_encode(dev(a, gpu), dev(b, gpu)).cpu().numpy()
    ref = co.box_encode(a.numpy(), b.numpy())
    assert H.max_rel_elem(enc, ref, 1e-3) < TOL
    d = torch.from_numpy(ref) * 0.3
    dec = ops.box_decode(dev(a, gpu), dev(d, gpu), 832.0, 832.0).cpu().numpy()
    rdec = co.box_decode(a.numpy(), d.numpy(), 832.0, 832.0)
    H.assert_elem_close(dec, rdec, TOL, "box_decode")


def test_ciou_fwd_bwd(gpu, ops):
    g = torch.Generator().manual_seed(9)
    N = 5000
    c = torch.rand((N, 2), generator=g) * 600 + 20
    wh = torch.rand((N, 2), generator=g) * 100 + 5
    p = torch.cat([c - wh / 2, c + wh / 2], 1)
    c2 = c + torch.randn((N, 2), generator=g) * 30
    wh2 = wh * torch.exp(torch.randn((N, 2), generator=g) * 0.3)
    t = torch.cat([c2 - wh2 / 2, c2 + wh2 / 2], 1)
    pd = dev(p, gpu).requires_grad_(True)
    out = ops.ciou(pd, dev(t, gpu))
    go = torch.rand((N,), generator=g)
    out.backward(dev(go, gpu))
    ref = co.ciou_fwd(p.numpy(), t.numpy())
    rgrad = co.ciou_bwd(p.numpy(), t.numpy(), go.numpy())
    assert np.abs(out.detach().cpu().numpy() - ref).max() < TOL
    H.assert_elem_close(pd.grad.cpu().numpy(), rgrad, TOL, "CIoU gradient")


# ------------------------------------------------------------------------------- NMS
@pytest.mark.parametrize("B,N,with_cls,max_out,max_cand", [(2, 500, False, 100, 0), (3, 3000, True, 300, 0),
                                                          (2, 6000, False, 1000, 4000), (1, 70, True, 70, 0),
                                                          (2, 9000, True, 9000, 0)])
def test_nms_keep_bit_exact(gpu, ops, B, N, with_cls, max_out, max_cand):
    boxes = H.clustered_boxes(B, N, 832, 11)
    scores = H.distinct_scores(B, N, 12)
    cls = torch.randint(0, 4, (B, N), generator=torch.Generator().manual_seed(13), dtype=torch.int32) if with_cls else None
    nv = torch.tensor([N, N - 17, N // 2][:B], dtype=torch.int32)
    keep, cnt = ops.nms_batched(dev(boxes, gpu), dev(scores, gpu), None if cls is None else dev(cls, gpu), dev(nv, gpu),
                                0.05, 0.5, max_cand, max_out)
    rk, rc = co.nms_batched(boxes.numpy(), scores.numpy(), None if cls is None else cls.numpy(), nv.numpy(), 0.05, 0.5,
                            max_cand, max_out)
    assert np.array_equal(cnt.cpu().numpy(), rc)
    assert np.array_equal(keep.cpu().numpy(), rk)


def test_nms_ties_and_empty(gpu, ops):
    B, N = 2, 400
    boxes = H.clustered_boxes(B, N, 640, 21)
    scores = (torch.randint(0, 8, (B, N), generator=torch.Generator().manual_seed(22)).float() / 8.0)  # heavy ties
    nv = torch.tensor([N, 0], dtype=torch.int32)  # second image empty
    keep, cnt = ops.nms_batched(dev(boxes, gpu), dev(scores, gpu), None, dev(nv, gpu), -1.0, 0.4, 0, N)
    rk, rc = co.nms_batched(boxes.numpy(), scores.numpy(), None, nv.numpy(), -1.0, 0.4, 0, N)
    assert np.array_equal(cnt.cpu().numpy(), rc)
    assert np.array_equal(keep.cpu().numpy(), rk)


def test_nms_idempotent_large(gpu, ops):
    """Size-independent property at a size the oracle would need minutes for: NMS(NMS(x)) == NMS(x)."""
    B, N = 4, 30000
    boxes = dev(H.clustered_boxes(B, N, 1280, 31, n_centers=400), gpu)
    scores = dev(H.distinct_scores(B, N, 32), gpu)
    keep, cnt = ops.nms_batched(boxes, scores, None, None, 0.0, 0.5, 0, N)
    for b in range(B):
        k = keep[b, : int(cnt[b])].long()
        assert (k >= 0).all()
        kb, ks = boxes[b, k][None].contiguous(), scores[b, k][None].contiguous()
        keep2, cnt2 = ops.nms_batched(kb, ks, None, None, 0.0, 0.5, 0, kb.shape[1])
        assert int(cnt2[0]) == kb.shape[1]
        assert torch.equal(keep2[0].long(), torch.arange(kb.shape[1], device=gpu))
        # kept scores are descending
        assert (ks[0, 1:] <= ks[0, :-1]).all()


# ------------------------------------------------------------------------------- RPN proposals
# 832 / pre 2000 / post 1000 is BASELINE config 3's own proposal stage (5 levels x 3 anchors, NA = 57 723 per image)
@pytest.mark.parametrize("size,A,pre,post", [(256, 3, 300, 100), (416, 3, 1000, 300), (832, 3, 2000, 1000)])
def test_rpn_proposals(gpu, ops, size, A, pre, post):
    B = 2
    strides = [4, 8, 16, 32, 64]
    hw = [(math.ceil(size / s), math.ceil(size / s)) for s in strides]
    base = np.stack([co.anchor_base([8.0 * s], [0.5, 1.0, 2.0]) for s in strides])
    anchors = co.anchor_grid([h for h, _ in hw], [w for _, w in hw], strides, base, 0.5)
    level_sizes = [h * w * A for h, w in hw]
    NA = anchors.shape[0]
    g = torch.Generator().manual_seed(41)
    logits = torch.randn((B, NA), generator=g)
    deltas = torch.randn((B, NA, 4), generator=g) * 0.2
    rois, sc, cnt, idx = ops.rpn_proposals(dev(logits, gpu), dev(deltas, gpu), dev(torch.from_numpy(anchors), gpu),
                                           level_sizes, pre, post, 0.7, 2.0, float(size), float(size), return_index=True)
    r_rois, r_sc, r_cnt, r_idx = co.rpn_proposals(logits.numpy(), deltas.numpy(), anchors, level_sizes, pre, post, 0.7, 2.0,
                                                  float(size), float(size), return_index=True)
    assert np.array_equal(cnt.cpu().numpy(), r_cnt)
    # the keep set, bit-exact: the anchor index of every surviving proposal, in output order, -1 padded
    # (north_star: "bit-exact for anchor indices / NMS keep masks")
    assert np.array_equal(idx.cpu().numpy(), r_idx)
    # decoded boxes go through expf (device libm vs glibc differ by ulps): tolerance on the coordinates only
    assert np.abs(rois.cpu().numpy() - r_rois).max() < 1e-2 * 1e-2 * size + 1e-3
    assert np.abs(sc.cpu().numpy() - r_sc).max() < TOL
    # without the index output the other outputs are unchanged
    rois2, sc2, cnt2 = ops.rpn_proposals(dev(logits, gpu), dev(deltas, gpu), dev(torch.from_numpy(anchors), gpu), level_sizes,
                                         pre, post, 0.7, 2.0, float(size), float(size))
    assert torch.equal(rois2, rois) and torch.equal(sc2, sc) and torch.equal(cnt2, cnt)


def test_rpn_proposals_golden_index(gpu, ops):
    gold = np.load(H.GOLDEN)
    rois, sc, cnt, idx = ops.rpn_proposals(dev(torch.from_numpy(gold["rpn_logits"]), gpu), dev(torch.from_numpy(gold["rpn_deltas"]), gpu),
                                           dev(torch.from_numpy(gold["anchor_grid"]), gpu), [432, 108, 27], 100, 40, 0.7, 2.0,
                                           96.0, 96.0, return_index=True)
    assert np.array_equal(cnt.cpu().numpy(), gold["rpn_count"])
    assert np.array_equal(idx.cpu().numpy(), gold["rpn_index"])


# ------------------------------------------------------------------------------- RoIAlign
def _roi_case(seed, B, C, size, R, strides):
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn((B, C, size // s, size // s), generator=g) for s in strides]
    bx = H.boxes_batch(1, R, size, seed + 1, 4.0, size * 0.8)[0]
    bi = torch.randint(0, B, (R, 1), generator=g).float()
    rois = torch.cat([bi, bx], 1)
    return feats, rois


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("aligned,sr", [(True, 2), (False, 0)])
def test_roi_align_fwd_bwd(gpu, ops, layout, aligned, sr):
    strides = [4, 8, 16, 32]
    feats, rois = _roi_case(51, 2, 48, 256, 60, strides)
    scales = [1.0 / s for s in strides]
    fd = [dev(f, gpu) for f in feats]
    if layout == "nhwc":
        fd = [f.contiguous(memory_format=torch.channels_last) for f in fd]
    fd = [f.requires_grad_(True) for f in fd]
    out = ops.roi_align(fd, dev(rois, gpu), 7, scales, sr, aligned, 2, 224.0 / 4, 4, channels_last_out=(layout == "nhwc"))
    ref = co.roi_align_fwd([f.numpy() for f in feats], rois.numpy(), 7, scales, sr, aligned, 2, 224.0 / 4, 4)
    assert out.shape == ref.shape
    H.assert_elem_close(out.detach().cpu().numpy(), ref, TOL, "RoIAlign forward")
    lv = ops.roi_level_map(dev(rois, gpu), 2, 4, 224.0 / 4, 4).cpu().numpy()
    assert np.array_equal(lv, co.roi_level_map(rois.numpy(), 2, 4, 224.0 / 4, 4))
    assert len(set(lv.tolist())) >= 3  # the case really spans levels
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(52))
    out.backward(dev(go, gpu))
    rg = co.roi_align_bwd([tuple(f.shape) for f in feats], rois.numpy(), go.numpy(), scales, sr, aligned, 2, 224.0 / 4, 4)
    for f, r in zip(fd, rg):
        H.assert_elem_close(f.grad.cpu().numpy(), r, TOL, "RoIAlign backward")


def _bf16_round(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("sr", [1, 2])
@pytest.mark.parametrize("in_dtype,out_dtype", [("f32", "f32"), ("bf16", "f32"), ("bf16", "bf16"), ("f32", "bf16")])
def test_roi_align_channels_last_fast_path(gpu, ops, sr, in_dtype, out_dtype):
    """The separable channels-last kernels (float32 or bfloat16 maps, float32 accumulation): the oracle is fed the
    same bf16-rounded features, so the float32-output cases hold the 1e-4 bar; a bf16 output adds its own rounding
    (2^-9 relative).  C = 64 so both vector widths (4 x f32, 8 x bf16) tile the channels; RoIs span all levels,
    touch the map borders and include sub-pixel and out-of-image boxes."""
    strides = [4, 8, 16, 32]
    feats, rois = _roi_case(53 + sr, 2, 64, 256, 90, strides)
    extra = torch.tensor([[0, -20.0, -20.0, 30.0, 30.0], [1, 200.0, 180.0, 300.0, 290.0], [0, 10.2, 10.4, 10.9, 11.1],
                          [1, 0.0, 0.0, 256.0, 256.0], [0, 255.0, 255.0, 256.0, 256.0], [1, -50.0, -50.0, -10.0, -10.0],
                          [0, 3.0, 100.0, 250.0, 104.0], [1, 100.0, 3.0, 104.0, 250.0]])
    rois = torch.cat([rois, extra])
    scales = [1.0 / s for s in strides]
    tin = torch.bfloat16 if in_dtype == "bf16" else torch.float32
    tout = torch.bfloat16 if out_dtype == "bf16" else torch.float32
    ref_feats = [_bf16_round(f) if in_dtype == "bf16" else f for f in feats]
    fd = [dev(f, gpu).to(tin).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in ref_feats]
    out = ops.roi_align(fd, dev(rois, gpu), 7, scales, sr, True, 2, 224.0 / 4, 4, channels_last_out=True, out_dtype=tout)
    assert out.dtype == tout and out.is_contiguous(memory_format=torch.channels_last)
    ref = co.roi_align_fwd([f.numpy() for f in ref_feats], rois.numpy(), 7, scales, sr, True, 2, 224.0 / 4, 4)
    tol = TOL if out_dtype == "f32" else 6e-3
    # element-wise (a max-norm bound would hide a wrong small bin): |out - ref| <= tol |ref| + tol median|ref|
    H.assert_elem_close(out.detach().float().cpu().numpy(), ref, tol, "RoIAlign channels-last forward")
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(57))
    go_ref = _bf16_round(go) if out_dtype == "bf16" else go
    out.backward(dev(go_ref, gpu).to(tout).contiguous(memory_format=torch.channels_last))
    rg = co.roi_align_bwd([tuple(f.shape) for f in feats], rois.numpy(), go_ref.numpy(), scales, sr, True, 2, 224.0 / 4, 4)
    for f, r in zip(fd, rg):
        assert f.grad.dtype == tin
        H.assert_elem_close(f.grad.float().cpu().numpy(), r, TOL if in_dtype == "f32" else 6e-3, "RoIAlign channels-last backward")


def test_roi_align_fast_path_equals_generic_path(gpu, ops):
    """Same inputs through the separable channels-last kernel and the any-stride kernel (NCHW maps): the two
    summation orders agree to float32 rounding, forward and backward."""
    strides = [4, 8, 16, 32]
    feats, rois = _roi_case(58, 3, 32, 320, 200, strides)
    scales = [1.0 / s for s in strides]
    a = [dev(f, gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
    b = [dev(f, gpu).contiguous().requires_grad_(True) for f in feats]
    oa = ops.roi_align(a, dev(rois, gpu), 7, scales, 2, True, 2, 224.0 / 4, 4, channels_last_out=True)
    ob = ops.roi_align(b, dev(rois, gpu), 7, scales, 2, True, 2, 224.0 / 4, 4, channels_last_out=False)
    assert H.rel_err(oa.detach().cpu().numpy(), ob.detach().cpu().numpy()) < 1e-5
    go = torch.randn(oa.shape, generator=torch.Generator().manual_seed(59)).to(gpu)
    oa.backward(go.contiguous(memory_format=torch.channels_last))
    ob.backward(go)
    for x, y in zip(a, b):
        assert H.rel_err(x.grad.cpu().numpy(), y.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_roi_pool_fwd_bwd(gpu, ops, layout):
    """README.md:73 "-RoIPooling" variant: values and argmax bit-exact, backward = scatter to argmax."""
    strides = [4, 8, 16, 32]
    feats, rois = _roi_case(61, 2, 48, 256, 60, strides)
    scales = [1.0 / s for s in strides]
    fd = [dev(f, gpu) for f in feats]
    if layout == "nhwc":
        fd = [f.contiguous(memory_format=torch.channels_last) for f in fd]
    fd = [f.requires_grad_(True) for f in fd]
    out, arg = ops.roi_pool(fd, dev(rois, gpu), 7, scales, 2, 224.0 / 4, 4, channels_last_out=(layout == "nhwc"),
                            return_argmax=True)
    ref, rarg = co.roi_pool_fwd([f.numpy() for f in feats], rois.numpy(), 7, scales, 2, 224.0 / 4, 4)
    assert np.array_equal(out.detach().cpu().numpy(), ref)
    assert np.array_equal(arg.cpu().numpy(), rarg)
    assert (rarg >= 0).mean() > 0.9
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(62))
    out.backward(dev(go, gpu))
    rg = co.roi_pool_bwd([tuple(f.shape) for f in feats], rois.numpy(), go.numpy(), rarg, scales, 2, 224.0 / 4, 4)
    for f, r in zip(fd, rg):
        assert H.rel_err(f.grad.cpu().numpy(), r) < TOL  # atomic summation order differs


def test_roi_pool_single_level_quantisation_edges(gpu, ops):
    """RoIs off the map, inverted, sub-cell and with .5 corners, on one stride-1 level."""
    g = torch.Generator().manual_seed(63)
    f = torch.randn((2, 16, 20, 24), generator=g)
    rois = torch.tensor([[0, 2.5, 2.5, 2.5, 2.5], [1, 20.0, 14.0, 60.0, 50.0], [0, 40.0, 40.0, 50.0, 50.0],
                         [1, 9.0, 9.0, 4.0, 4.0], [0, -6.0, -3.0, 5.5, 4.5], [1, 0.0, 0.0, 23.0, 19.0],
                         [0, 3.49, 7.51, 3.51, 7.49]])
    for P in (1, 3, 7):
        out, arg = ops.roi_pool([dev(f, gpu)], dev(rois, gpu), P, [1.0], 2, 224.0, 4, return_argmax=True)
        ref, rarg = co.roi_pool_fwd([f.numpy()], rois.numpy(), P, [1.0], 2, 224.0, 4)
        assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(arg.cpu().numpy(), rarg)
    assert (rarg[2] == -1).all()


# ------------------------------------------------------------------------------- YOLOv5
@pytest.mark.parametrize("no,size,B", [(85, 128, 2), (9, 96, 3)])
def test_yolo_decode(gpu, ops, no, size, B):
    preds = H.yolo_preds(B, size, no, 61)
    out = ops.yolo_decode([dev(p, gpu) for p in preds], H.YOLO_STRIDES, H.YOLO_ANCHORS).cpu().numpy()
    ref = co.yolo_decode([p.numpy() for p in preds], H.YOLO_STRIDES, H.YOLO_ANCHORS)
    assert H.max_rel_elem(out, ref, 1e-4) < TOL


def test_yolo_decode_channels_last_view(gpu, ops):
    """The conv output [B, na*no, H, W] in channels-last memory, viewed as [B,na,H,W,no] without a copy."""
    B, no, size = 2, 85, 128
    preds = H.yolo_preds(B, size, no, 62)
    views = []
    for p in preds:
        _, na, Hh, Ww, _ = p.shape
        conv = p.permute(0, 1, 4, 2, 3).reshape(B, na * no, Hh, Ww).to(gpu).contiguous(memory_format=torch.channels_last)
        v = conv.view(B, na, no, Hh, Ww).permute(0, 1, 3, 4, 2)
        assert v.stride(4) == 1 and v.data_ptr() == conv.data_ptr()
        views.append(v)
    out = ops.yolo_decode(views, H.YOLO_STRIDES, H.YOLO_ANCHORS).cpu().numpy()
    ref = co.yolo_decode([p.numpy() for p in preds], H.YOLO_STRIDES, H.YOLO_ANCHORS)
    assert H.max_rel_elem(out, ref, 1e-4) < TOL


def _yolo_targets(B, size, nc, seed, style="coco", m_pad=40):
    from heltondetection_amd import synthetic as syn
    g = syn.make_generator(seed)
    return syn.make_targets(B, size, g, style, nc, m_pad)


@pytest.mark.parametrize("size,B,style,m_pad", [(128, 2, "coco", 40), (256, 2, "visdrone", 600)])
def test_yolo_assign_bit_exact(gpu, ops, size, B, style, m_pad):
    gtb, gtc, mv = _yolo_targets(B, size, 80, 71, style, m_pad)
    shapes = H.yolo_shapes(size, 85)
    cell, tbox, cls = ops.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, dev(gtb, gpu), dev(gtc, gpu), dev(mv, gpu), 4.0)
    rc, rt, rl = co.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, gtb.numpy(), gtc.numpy(), mv.numpy(), 4.0)
    assert (rc >= 0).sum() > 0
    assert np.array_equal(cell.cpu().numpy(), rc)
    assert np.array_equal(tbox.cpu().numpy(), rt)
    assert np.array_equal(cls.cpu().numpy(), rl)


FOCAL_CASES = [None, (1.5, 0.25, ("cls", "obj")), (0.5, -1.0, ("cls",)), (0.5, 0.25, ("obj",))]


@pytest.mark.parametrize("no,layout,focal", [(85, "dense", None), (85, "channels_last", None), (6, "dense", None),
                                             (85, "channels_last", FOCAL_CASES[1]), (85, "dense", FOCAL_CASES[2]),
                                             (6, "channels_last", FOCAL_CASES[3])])
def test_yolo_loss_fwd_bwd(gpu, ops, no, layout, focal):
    """focal != None: the "YOLOv5s-focalloss(...)" variants (README.md:118-154); gamma 0.5 is a
    "root"-style modulation, terms=("cls",) wraps the classification BCE only."""
    B, size = 2, 128
    nc = no - 5
    preds = H.yolo_preds(B, size, no, 81)
    gtb, gtc, mv = _yolo_targets(B, size, max(nc, 1), 82, "coco", 24)
    shapes = H.yolo_shapes(size, no)
    rc, rt, rl = co.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, gtb.numpy(), gtc.numpy(), mv.numpy(), 4.0)
    r_loss, r_grads, _, _ = co.yolo_loss([p.numpy() for p in preds], H.YOLO_STRIDES, H.YOLO_ANCHORS, rc, rt, rl, 24,
                                         grad_scale=0.5, focal=focal)
    slots = ops.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, dev(gtb, gpu), dev(gtc, gpu), dev(mv, gpu), 4.0)
    pd = []
    for p in preds:
        if layout == "dense":
            pd.append(dev(p, gpu).requires_grad_(True))
        else:
            _, na, Hh, Ww, _ = p.shape
            conv = p.permute(0, 1, 4, 2, 3).reshape(B, na * no, Hh, Ww).to(gpu).contiguous(memory_format=torch.channels_last)
            conv.requires_grad_(True)
            pd.append(conv)
    views = pd if layout == "dense" else [c.view(B, 3, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2) for c in pd]
    biases = [torch.zeros(3 * no, device=gpu, requires_grad=True) for _ in preds]  # values already in preds
    total, parts = ops.yolo_loss(views, H.YOLO_STRIDES, H.YOLO_ANCHORS, slots, 24, biases=biases, focal=focal)
    (total * 0.5).backward()
    assert H.max_rel_elem(parts.cpu().numpy(), r_loss, 1e-6) < TOL
    for l, p in enumerate(pd):
        g = p.grad
        if layout != "dense":
            g = g.view(B, 3, no, g.shape[2], g.shape[3]).permute(0, 1, 3, 4, 2)
        assert H.rel_err(g.cpu().numpy(), r_grads[l]) < TOL
        # bias gradient = sum of the dense gradient over (b, y, x), per (anchor, channel)
        ref_b = r_grads[l].astype(np.float64).sum(axis=(0, 2, 3)).reshape(-1)
        assert H.rel_err(biases[l].grad.cpu().numpy(), ref_b) < TOL


def test_yolo_nms(gpu, ops):
    B, size, no = 2, 160, 85
    preds = H.yolo_preds(B, size, no, 91, scale=2.0)
    dec = co.yolo_decode([p.numpy() for p in preds], H.YOLO_STRIDES, H.YOLO_ANCHORS)
    det, cnt = ops.yolo_nms(dev(torch.from_numpy(dec), gpu), 0.3, 0.45, 2000, 100)
    rd, rc = co.yolo_nms(dec, 0.3, 0.45, 2000, 100)
    assert rc.min() > 5
    assert np.array_equal(cnt.cpu().numpy(), rc)
    assert np.array_equal(det.cpu().numpy(), rd)


@pytest.mark.parametrize("layout", ["dense", "channels_last"])
@pytest.mark.parametrize("conf", [0.3, 0.02])
def test_yolo_detect_equals_decode_then_nms(gpu, ops, layout, conf):
    """hd_yolo_detect (candidate pass on the raw head outputs) must give exactly the detections of
    hd_yolo_decode + hd_yolo_nms and of the oracle: same candidates, same keys, same ties."""
    B, size, no = 2, 160, 85
    preds = H.yolo_preds(B, size, no, 93, scale=2.0)
    dec = co.yolo_decode([p.numpy() for p in preds], H.YOLO_STRIDES, H.YOLO_ANCHORS)
    rd, rc = co.yolo_nms(dec, conf, 0.45, 3000, 100)
    assert rc.min() > 3
    if layout == "dense":
        views = [dev(p, gpu) for p in preds]
    else:
        views = []
        for p in preds:
            _, na, Hh, Ww, _ = p.shape
            conv = p.permute(0, 1, 4, 2, 3).reshape(B, na * no, Hh, Ww).to(gpu).contiguous(memory_format=torch.channels_last)
            views.append(conv.view(B, na, no, Hh, Ww).permute(0, 1, 3, 4, 2))
    det, cnt = ops.yolo_detect(views, H.YOLO_STRIDES, H.YOLO_ANCHORS, conf, 0.45, 3000, 100)
    d2, c2 = ops.yolo_nms(ops.yolo_decode(views, H.YOLO_STRIDES, H.YOLO_ANCHORS), conf, 0.45, 3000, 100)
    assert torch.equal(cnt, c2) and torch.equal(det, d2)          # bit-identical to the two-step HIP path
    # vs the oracle: same detections; values within the decode tolerance (device expf vs libm differ by ulps)
    d = det.cpu().numpy()
    assert np.array_equal(cnt.cpu().numpy(), rc) and np.array_equal(d[..., 5], rd[..., 5])
    assert np.allclose(d[..., :5], rd[..., :5], rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------- WBF
def test_wbf(gpu, ops):
    B, K, T = 3, 600, 3
    g = torch.Generator().manual_seed(101)
    base = H.clustered_boxes(B, K // T, 640, 102, n_centers=60, jitter=3.0)
    boxes = torch.cat([base + torch.randn(base.shape, generator=g) * 2.0 for _ in range(T)], 1).clamp(0, 640)
    scores = torch.rand((B, K), generator=g)
    labels = torch.randint(0, 5, (B, K // T), generator=g, dtype=torch.int32).repeat(1, T)
    mw = torch.cat([torch.full((B, K // T), w) for w in (1.0, 1.0, 0.5)], 1)
    kv = torch.tensor([K, K - 50, 10], dtype=torch.int32)
    out, cnt = ops.wbf(dev(boxes, gpu), dev(scores, gpu), dev(labels, gpu), dev(mw, gpu), dev(kv, gpu), 2.5, 0.55, 0.05)
    ro, rc = co.wbf(boxes.numpy(), scores.numpy(), labels.numpy(), mw.numpy(), kv.numpy(), 2.5, 0.55, 0.05)
    assert np.array_equal(cnt.cpu().numpy(), rc)
    assert np.array_equal(out.cpu().numpy(), ro)


def test_wbf_config5_density(gpu, ops):
    """BASELINE config 5 (FasterRCNN-PAFPN 1280^2 VisDrone-style, README.md:105): 3 TTA views x ~500 objects of 10 classes
    per image = K 1500 candidates, B = 8 — the bench micro shape — bit-exact against the oracle, ragged valid counts."""
    B, per, T = 8, 500, 3
    K = per * T
    g = torch.Generator().manual_seed(131)
    from heltondetection_amd import synthetic as syn
    base = torch.stack([syn.random_boxes(per, 1280.0, g, 4.0, 64.0) for _ in range(B)])
    boxes = torch.cat([base + torch.randn(base.shape, generator=g) * 1.5 for _ in range(T)], 1).clamp(0, 1280.0)
    scores = torch.rand((B, K), generator=g) * 0.7 + 0.3
    labels = torch.randint(0, 10, (B, per), generator=g, dtype=torch.int32).repeat(1, T)
    mw = torch.cat([torch.full((B, per), w) for w in (1.0, 1.0, 0.5)], 1)
    kv = torch.tensor([K, K, K - 137, K, 1200, K, 900, K], dtype=torch.int32)
    out, cnt = ops.wbf(dev(boxes, gpu), dev(scores, gpu), dev(labels, gpu), dev(mw, gpu), dev(kv, gpu), 3.0, 0.55, 0.05)
    ro, rc = co.wbf(boxes.numpy(), scores.numpy(), labels.numpy(), mw.numpy(), kv.numpy(), 3.0, 0.55, 0.05)
    assert np.array_equal(cnt.cpu().numpy(), rc)
    assert int(rc.min()) > 300  # dense: hundreds of fused boxes per image
    assert np.array_equal(out.cpu().numpy(), ro)


# ------------------------------------------------------------------------------- SPPF max-pool
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool5_matches_torch(gpu, ops, dtype):
    """Checked against PyTorch's own max_pool2d on the CPU (a torch fp32 reference is the right checker
    for this backbone glue op; values are distinct so arg-max ties cannot differ)."""
    g = torch.Generator().manual_seed(111)
    B, C, Hh, Ww = 3, 64, 13, 20
    x = torch.randn((B, C, Hh, Ww), generator=g)
    x = (x + torch.arange(x.numel()).view_as(x) * 1e-3).to(dtype).float()   # distinct after rounding
    xd = x.to(gpu).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.maxpool5(xd)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(xr, 5, 1, 2)
    assert torch.equal(y.float().cpu(), yr.detach())
    go = torch.randn(y.shape, generator=g).to(dtype).float()
    y.backward(go.to(gpu).to(dtype).contiguous(memory_format=torch.channels_last))
    yr.backward(go)
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert H.rel_err(xd.grad.float().cpu().numpy(), xr.grad.numpy()) < tol


# ------------------------------------------------------------------------------- fused BN + act
@pytest.mark.parametrize("dtype,act,C", [(torch.float32, "silu", 32), (torch.bfloat16, "silu", 64),
                                          (torch.float32, "relu", 48), (torch.bfloat16, None, 256)])
def test_bn_act_matches_torch(gpu, ops, dtype, act, C):
    """Backbone glue: checked against PyTorch's BatchNorm2d + activation in float64 on the CPU."""
    g = torch.Generator().manual_seed(121)
    B, Hh, Ww = 4, 9, 11
    x = (torch.randn((B, C, Hh, Ww), generator=g) * 1.5 + 0.3).to(dtype)
    w = torch.rand(C, generator=g) + 0.5
    b = torch.randn(C, generator=g) * 0.2
    rm, rv = torch.zeros(C), torch.ones(C)
    go = torch.randn((B, C, Hh, Ww), generator=g).to(dtype)
    # reference (float64)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    rmr, rvr = rm.double().clone(), rv.double().clone()
    z = torch.nn.functional.batch_norm(xr, rmr, rvr, wr, br, True, 0.03, 1e-3)
    yr = {"silu": torch.nn.functional.silu, "relu": torch.relu, None: (lambda t: t)}[act](z)
    yr.backward(go.double())
    # HIP
    xd = x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(gpu).requires_grad_(True), b.to(gpu).requires_grad_(True)
    rmd, rvd = rm.to(gpu), rv.to(gpu)
    ops.bn_arena_reset(gpu)
    y = ops.bn_act(xd, wd, bd, rmd, rvd, 1e-3, 0.03, act, True)
    y.backward(go.to(gpu).contiguous(memory_format=torch.channels_last))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert H.rel_err(y.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    assert H.rel_err(xd.grad.float().cpu().numpy(), xr.grad.numpy()) < tol
    assert H.rel_err(wd.grad.cpu().numpy(), wr.grad.numpy()) < tol
    assert H.rel_err(bd.grad.cpu().numpy(), br.grad.numpy()) < tol
    assert H.rel_err(rmd.cpu().numpy(), rmr.numpy()) < 1e-4 and H.rel_err(rvd.cpu().numpy(), rvr.numpy()) < 1e-4
    # second pass without an arena reset must still be right (slot self-clears), and inference mode
    y2 = ops.bn_act(xd.detach(), wd, bd, rmd.clone(), rvd.clone(), 1e-3, 0.03, act, True)
    assert H.rel_err(y2.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    yi = ops.bn_act(xd.detach(), wd, bd, rmd, rvd, 1e-3, 0.03, act, False)
    zi = torch.nn.functional.batch_norm(x.double(), rmd.cpu().double(), rvd.cpu().double(), w.double(), b.double(), False, 0.03, 1e-3)
    yir = {"silu": torch.nn.functional.silu, "relu": torch.relu, None: (lambda t: t)}[act](zi)
    assert H.rel_err(yi.float().cpu().numpy(), yir.numpy()) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_act_backward_reads_cat_gradient_slices_in_place(gpu, ops, dtype):
    """Two fused BN+SiLU outputs go through torch.cat (as in every C3 block): their backward passes receive
    channel slices of the concatenation's gradient (row stride 2C) and must read them in place — same
    gradients as the float64 reference, and no copy of the slice (ops._cl_row_stride recognises it)."""
    g = torch.Generator().manual_seed(131)
    B, C, Hh, Ww = 3, 32, 7, 10
    xs = [(torch.randn((B, C, Hh, Ww), generator=g) * 1.2).to(dtype) for _ in range(2)]
    ws = [torch.rand(C, generator=g) + 0.5 for _ in range(2)]
    bs = [torch.randn(C, generator=g) * 0.2 for _ in range(2)]
    go = torch.randn((B, 2 * C, Hh, Ww), generator=g).to(dtype)
    # reference
    xr = [x.double().requires_grad_(True) for x in xs]
    wr = [w.double().requires_grad_(True) for w in ws]
    br = [b.double().requires_grad_(True) for b in bs]
    yr = torch.cat([torch.nn.functional.silu(torch.nn.functional.batch_norm(x, None, None, w, b, True, 0.03, 1e-3))
                    for x, w, b in zip(xr, wr, br)], 1)
    yr.backward(go.double())
    # HIP
    xd = [x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True) for x in xs]
    wd = [w.to(gpu).requires_grad_(True) for w in ws]
    bd = [b.to(gpu).requires_grad_(True) for b in bs]
    ops.bn_arena_reset(gpu)
    y = torch.cat([ops.bn_act(x, w, b, None, None, 1e-3, 0.03, "silu", True) for x, w, b in zip(xd, wd, bd)], 1)
    god = go.to(gpu).contiguous(memory_format=torch.channels_last)
    sl = god[:, C:]
    vec = 4 if dtype == torch.float32 else 8
    assert ops._cl_row_stride(sl, vec) == 2 * C and ops._cl_row_stride(god, vec) == 2 * C
    assert ops._cl_row_stride(god[:, 4:4 + C], 8) is None or dtype == torch.float32  # misaligned for 16-byte bf16 vectors
    y.backward(god)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert H.rel_err(y.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    for k in range(2):
        assert H.rel_err(xd[k].grad.float().cpu().numpy(), xr[k].grad.numpy()) < tol
        assert H.rel_err(wd[k].grad.cpu().numpy(), wr[k].grad.numpy()) < tol
        assert H.rel_err(bd[k].grad.cpu().numpy(), br[k].grad.numpy()) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act,after", [("relu", False), ("silu", True), ("silu", False), (None, False)])
def test_bn_act_with_residual_matches_torch(gpu, ops, dtype, act, after):
    """Residual input of the fused BatchNorm: act(bn(x) + r) (ResNet bottleneck tail) and act(bn(x)) + r (YOLOv5
    shortcut), forward and all four gradients against float64 PyTorch; inference mode computes the same."""
    g = torch.Generator().manual_seed(161)
    B, C, Hh, Ww = 3, 16, 8, 9
    x = (torch.randn((B, C, Hh, Ww), generator=g) * 1.4 + 0.1).to(dtype)
    r = torch.randn((B, C, Hh, Ww), generator=g).to(dtype)
    w = torch.rand(C, generator=g) + 0.5
    b = torch.randn(C, generator=g) * 0.3
    go = torch.randn((B, C, Hh, Ww), generator=g).to(dtype)
    f = {"silu": torch.nn.functional.silu, "relu": torch.relu, None: (lambda t: t)}[act]
    xr, rr = x.double().requires_grad_(True), r.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    z = torch.nn.functional.batch_norm(xr, None, None, wr, br, True, 0.03, 1e-3)
    yr = f(z) + rr if after else f(z + rr)
    yr.backward(go.double())
    cl = lambda t: t.to(gpu).contiguous(memory_format=torch.channels_last)
    xd, rd = cl(x).requires_grad_(True), cl(r).requires_grad_(True)
    wd, bd = w.to(gpu).requires_grad_(True), b.to(gpu).requires_grad_(True)
    rm, rv = torch.zeros(C, device=gpu), torch.ones(C, device=gpu)
    ops.bn_arena_reset(gpu)
    y = ops.bn_act(xd, wd, bd, rm, rv, 1e-3, 0.03, act, True, residual=rd, residual_after_act=after)
    y.backward(cl(go))
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    assert H.rel_err(y.detach().float().cpu().numpy(), yr.detach().numpy()) < tol
    for got, ref in ((xd.grad, xr.grad), (rd.grad, rr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        assert H.rel_err(got.float().cpu().numpy(), ref.numpy()) < tol
    # inference mode: running statistics, same expression
    yi = ops.bn_act(xd.detach(), wd, bd, rm, rv, 1e-3, 0.03, act, False, residual=rd.detach(), residual_after_act=after)
    zi = torch.nn.functional.batch_norm(x.double(), rm.cpu().double(), rv.cpu().double(), w.double(), b.double(), False, 0.03, 1e-3)
    yir = f(zi) + r.double() if after else f(zi + r.double())
    assert H.rel_err(yi.float().cpu().numpy(), yir.numpy()) < tol


# ------------------------------------------------------------------------------- hipGraph memset rewrite
def test_graph_memset_nodes_are_rewritten(gpu, ops):
    """Capture a small graph containing real hipMemsetAsync calls (1-byte pattern on an odd length, 4-byte
    aligned span) between kernels, rewrite the memset nodes, instantiate, replay: same results, no memset left."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    a = torch.full((1001,), 7, dtype=torch.uint8, device=gpu)
    b = torch.ones((4096,), dtype=torch.float32, device=gpu)
    x = torch.arange(64, dtype=torch.float32, device=gpu)
    y = torch.zeros_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y.copy_(x * 2)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        st = torch.cuda.current_stream().cuda_stream
        y.copy_(x * 2)
        assert hip.hipMemsetAsync(a.data_ptr() + 3, 0x5A, 997, st) == 0      # unaligned, odd length
        assert hip.hipMemsetAsync(b.data_ptr(), 0, 4096 * 4, st) == 0         # aligned words
        y.add_(b[:64])                                                        # depends on the memset of b
    nodes, replaced = ops.graph_replace_memsets(g.raw_cuda_graph())
    assert replaced == 2 and nodes >= 4
    g.instantiate()
    for _ in range(3):
        a.fill_(7)
        b.fill_(1.0)
        g.replay()
        torch.cuda.synchronize()
        ar = a.cpu().numpy()
        assert (ar[:3] == 7).all() and (ar[3:1000] == 0x5A).all() and ar[1000] == 7
        assert float(b.abs().sum()) == 0.0
        assert torch.equal(y.cpu(), torch.arange(64, dtype=torch.float32) * 2)   # b was already zero when added
    nodes2, replaced2 = ops.graph_replace_memsets(g.raw_cuda_graph())
    assert replaced2 == 0


# ------------------------------------------------------------------------------- Mosaic-4
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_mosaic4_matches_oracle(gpu, ops, layout):
    """Pixels, boxes, classes and counts bit-exact (integer tile offsets, fixed summation order)."""
    from heltondetection_amd import synthetic as syn
    g = syn.make_generator(141)
    B, S, M = 6, 64, 9
    img = syn.make_images(B, S, g)
    gtb, gtc, mv = syn.make_targets(B, S, g, "coco", 7, M)
    src = torch.stack([torch.tensor([b] + torch.randint(0, B, (3,), generator=g).tolist()) for b in range(B)]).to(torch.int32)
    center = torch.randint(S // 2, 3 * S // 2 + 1, (B, 2), generator=g, dtype=torch.int32)
    center[0] = torch.tensor([S // 2, 3 * S // 2])      # extreme corners of the allowed range
    center[1] = torch.tensor([3 * S // 2, S // 2])
    apply = torch.tensor([1, 1, 0, 1, 1, 1], dtype=torch.int32)
    ref = co.mosaic4(img.numpy(), gtb.numpy(), gtc.numpy(), mv.numpy(), src.numpy(), center.numpy(), apply.numpy(),
                     min_wh=1.0)
    imd = dev(img, gpu)
    if layout == "nhwc":
        imd = imd.contiguous(memory_format=torch.channels_last)
    out, ob, oc, cnt = ops.mosaic4(imd, dev(gtb, gpu), dev(gtc, gpu), dev(mv, gpu), dev(src, gpu), dev(center, gpu),
                                   dev(apply, gpu), min_wh=1.0)
    assert out.stride() == imd.stride()
    assert np.array_equal(out.cpu().numpy(), ref[0])
    assert np.array_equal(cnt.cpu().numpy(), ref[3]) and ref[3].max() > M          # more boxes than one image had
    assert np.array_equal(ob.cpu().numpy(), ref[1]) and np.array_equal(oc.cpu().numpy(), ref[2])
    # pass-through image is untouched; mosaic boxes stay inside the image
    assert torch.equal(out[2], imd[int(src[2, 0])])
    v = ob[apply.bool().to(gpu)]
    assert float(v.min()) >= 0.0 and float(v.max()) <= S


def test_mosaic_module_draws_on_device_and_feeds_the_assigner(gpu, ops):
    """augment.Mosaic: device-side draws (reproducible with a device generator), p=0 is the identity on images and
    boxes, p=1 changes every image, and its padded output is consumed by the YOLOv5 assigner unchanged."""
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.augment import Mosaic
    g = syn.make_generator(142)
    B, S, M = 8, 128, 12
    img = dev(syn.make_images(B, S, g), gpu).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = [dev(t, gpu) for t in syn.make_targets(B, S, g, "coco", 8, M)]
    ident = Mosaic(p=0.0)(img, gtb, gtc, mv)
    assert torch.equal(ident[0], img) and torch.equal(ident[3], mv)
    for b in range(B):
        n = int(mv[b])
        assert torch.equal(ident[1][b, :n], gtb[b, :n]) and torch.equal(ident[2][b, :n], gtc[b, :n])
    outs = []
    for _ in range(2):
        dg = torch.Generator(device=gpu).manual_seed(7)
        outs.append(Mosaic(p=1.0, max_boxes=2 * M, generator=dg)(img, gtb, gtc, mv))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))               # same generator state, same batch
    o_img, o_box, o_cls, o_cnt = outs[0]
    assert o_box.shape == (B, 2 * M, 4) and int(o_cnt.max()) <= 2 * M
    assert all(not torch.equal(o_img[b], img[b]) for b in range(B))
    src, center, apply = Mosaic(p=0.5, generator=torch.Generator(device=gpu).manual_seed(3)).draw(64, S, gpu)
    assert bool((src[:, 0] == torch.arange(64, device=gpu)).all()) and bool((src[:, 1:] != src[:, :1]).all())
    assert int(center.min()) >= S // 2 and int(center.max()) <= 3 * S // 2 and 8 < int(apply.sum()) < 56
    shapes = H.yolo_shapes(S, 13)
    cell, tbox, cls = ops.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, o_box, o_cls, o_cnt, 4.0)
    assert int((cell >= 0).sum()) > 0


# ------------------------------------------------------------------------------- MaxIoUAssigner (host logic over hd_iou_assign)
def test_max_iou_assigner_low_quality_match_is_deterministic(gpu, ops):
    """padded gts used to be scattered onto box 0, colliding with a valid gt whose best box is box 0.
    Reference here: the sequential per-gt loop of the public MaxIoUAssigner on the oracle's IoU matrix."""
    from heltondetection_amd.assigners import MaxIoUAssigner
    B, N, M = 3, 400, 12
    a = H.boxes_batch(1, N, 256, 71)[0]
    b = H.boxes_batch(B, M, 256, 72)
    b[0, 0] = a[0]              # gt 0 of image 0 matches box 0 exactly -> its best box IS index 0
    b[1, 3] = b[1, 2]           # two gts sharing one best box: the later gt wins the forced match
    mv = torch.tensor([5, 12, 0], dtype=torch.int32)   # images with padded gts / none at all
    asg = MaxIoUAssigner(0.7, 0.3, True)
    outs = [asg(a.to(gpu), b.to(gpu), mv.to(gpu)) for _ in range(3)]
    for o in outs[1:]:
        assert all(torch.equal(x, y) for x, y in zip(o, outs[0]))
    labels, matched, mx = [t.cpu().numpy() for t in outs[0]]
    iou = co.pairwise_iou(np.broadcast_to(a.numpy(), (B, N, 4)).copy(), b.numpy())
    for i in range(B):
        m = int(mv[i])
        if m == 0:
            assert (labels[i] == -1).all() or (labels[i] <= 0).all()
            continue
        sub = iou[i, :, :m]
        r_mx, r_am = sub.max(1), sub.argmax(1)
        r_lab = np.full(N, -1, np.int32)
        r_lab[r_mx < 0.3] = 0
        r_lab[r_mx >= 0.7] = 1
        r_match = r_am.astype(np.int32)
        for j in range(m):                      # sequential: a later gt overwrites an earlier one
            if sub[:, j].max() > 0:
                n_best = int(sub[:, j].argmax())
                r_lab[n_best] = 1
                r_match[n_best] = j
        assert np.array_equal(labels[i], r_lab), i
        assert np.array_equal(matched[i], r_match), i
    assert labels[0, 0] == 1 and matched[0, 0] == 0


# ------------------------------------------------------------------------------- division-free quotient, on the device
def _adversarial_boxes(n, seed):
    """Boxes that steer the IoU kernels through every branch of the fast/IEEE quotient selection: ordinary boxes,
    the all-ones-denominator family (union = (2 - 2^-23) * 2^k with a power-of-two intersection, the one case the
    fast sequence may miss by an ulp and must hand to the IEEE division), and boxes the range check must refuse
    (-0.0, 1e-30, 1e30, inverted, zero-area)."""
    g = torch.Generator().manual_seed(seed)
    base = H.boxes_batch(1, n, 512, seed)[0]
    special = []
    eps = 2.0 ** -23
    for k in (-8, -3, 0, 1, 5, 9):
        s = 2.0 ** k
        special += [[0.0, 0.0, s, s], [(1 - eps) * s, 0.0, (2 - eps) * s, s]]          # inter = 2^-23 s^2, union all-ones
    special += [[-0.0, 0.0, 10.0, 10.0], [0.0, -0.0, 8.0, 12.0], [1e-30, 1e-30, 5.0, 5.0], [0.0, 0.0, 1e30, 1e30],
                [7.0, 7.0, 3.0, 3.0], [4.0, 4.0, 4.0, 9.0], [0.0, 0.0, 0.0, 0.0], [2.0, 2.0, 6.0, 6.0],
                [0.0, 0.0, float("inf"), 4.0]]
    sp = torch.tensor(special, dtype=torch.float32)
    idx = torch.randperm(n, generator=g)[: len(sp)]
    base[idx] = sp
    return base


@pytest.mark.parametrize("M", [24, 64, 500])
def test_iou_fast_quotient_paths_are_bit_exact(gpu, ops, M):
    """M = 24: flat kernel; 64 / 500: lane-owns-columns kernel (C = 4 with one and two spans).  Every tile flavour:
    all-safe tiles (packed fast quotient), tiles with an all-ones denominator (row recomputed), tiles with a
    refused box (IEEE division).  Bit-exact against the oracle, which divides."""
    B, N = 2, 1536
    a = torch.stack([_adversarial_boxes(N, 81 + i) for i in range(B)])
    b = torch.stack([_adversarial_boxes(M, 91 + i) for i in range(B)])
    # rows 0..127 of image 0 and all of its columns stay ordinary, so fully "safe" tiles exist as well
    a[0, :128] = H.boxes_batch(1, 128, 512, 97)[0]
    b[0] = H.boxes_batch(1, M, 512, 98)[0]
    # and a safe tile that contains ONLY the all-ones family in one row/column
    eps = 2.0 ** -23
    a[0, 5] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    b[0, 7] = torch.tensor([1 - eps, 0.0, 2 - eps, 1.0])
    got = ops.pairwise_iou(a.to(gpu), b.to(gpu)).cpu().numpy()
    ref = co.pairwise_iou(a.numpy(), b.numpy())
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert got[0, 5, 7] == np.float32(2.0 ** -23) / np.float32(2 - eps) and got[0, 5, 7] > 2.0 ** -24
    mv = torch.tensor([M, max(M - 5, 1)], dtype=torch.int32)
    for x, y in zip(ops.iou_assign(a.to(gpu), b.to(gpu), mv.to(gpu)), co.iou_assign(a.numpy(), b.numpy(), mv.numpy())):
        assert np.array_equal(x.cpu().numpy(), y)
    # shared anchors (a_bstride = 0)
    got = ops.pairwise_iou(a[0].to(gpu), b.to(gpu)).cpu().numpy()
    ref = co.pairwise_iou(np.broadcast_to(a[0].numpy(), (B, N, 4)).copy(), b.numpy())
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("C,R,dtype", [(256, 2048, "bf16"), (64, 300, "f32"), (320, 1500, "bf16")])
def test_roi_align_backward_gather_is_reproducible_and_equals_the_scatter(gpu, ops, C, R, dtype):
    """the backward as a gather (one workgroup per 8 x 8 tile of cells, RoIs in list order, no atomics) writes
    every cell of the gradient maps once in the maps' own dtype.  Checked: two runs are bit-identical (the scatter's float
    atomics are not), the result equals the float-atomic scatter to float32 rounding (bf16 maps: one bf16 rounding), cells
    under no RoI are exactly zero although the maps start uninitialised, R spans more than one scan chunk (1024), C is not a
    multiple of 256 in one case."""
    strides = [4, 8, 16, 32]
    B, size = 3, 416
    feats, rois = _roi_case(71 + C, B, C, size, R, strides)
    # leave image 2 without RoIs on its upper half: whole tiles that nothing touches
    rois[rois[:, 0] == 2, 2] = rois[rois[:, 0] == 2, 2].clamp(min=size * 0.6)
    rois[rois[:, 0] == 2, 4] = torch.maximum(rois[rois[:, 0] == 2, 4], rois[rois[:, 0] == 2, 2] + 8.0)
    if C == 256:   # a list ordered by image (what the RoI head produces): every tile scans its own image's range only
        rois = rois[torch.sort(rois[:, 0], stable=True).indices]
    scales = [1.0 / s for s in strides]
    t = torch.bfloat16 if dtype == "bf16" else torch.float32
    go = None
    res = {}
    for mode in ("gather", "gather2", "scatter"):
        ops.RoiBwdSwitch.gather = mode != "scatter"
        try:
            fd = [dev(f, gpu).to(t).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
            out = ops.roi_align(fd, dev(rois, gpu), 7, scales, 2, True, 2, 224.0 / 4, 4, channels_last_out=True)
            if go is None:
                go = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(gpu).to(t).contiguous(memory_format=torch.channels_last)
            out.backward(go)
            res[mode] = [f.grad.clone() for f in fd]
        finally:
            ops.RoiBwdSwitch.gather = True
    for a, b, c in zip(res["gather"], res["gather2"], res["scatter"]):
        assert a.dtype == t and torch.equal(a, b)                         # bit-reproducible
        assert bool(torch.isfinite(a).all())
        tol = 2e-5 if dtype == "f32" else 2.0 ** -7
        err = (a.float() - c.float()).abs()
        assert bool((err <= tol * c.float().abs() + tol * c.float().abs().mean()).all()), float(err.max())
        assert bool((a[c == 0] == 0).all())                               # untouched cells are written as zeros
    assert any(bool((g[2, :, : g.shape[2] // 4] == 0).all()) for g in res["gather"])   # the empty region exists


# ---- the DEFAULT training path against the oracle, directly -------------------------------------------------------------------
# Training with bf16 weight shadows runs hd_head_conv_fwd (float32 rows padded to a pitch of 32 / 64 / 128 / 256) ->
# hd_yolo_loss_fwd on the padded view -> hd_yolo_loss_bwd_dense_bf16 / _pos_bf16 (bf16 padded gradient rows, packed bf16 atomics
# for cells matched more than once) -> hd_head_conv_bwd.  To compare THAT chain with co.yolo_loss on bit-identical logits, the
# Detect convolution is given selection weights: every output channel n reads input channel(s) sel(n) with weight 1, so
#   logits[n] = sum of the selected bf16 inputs (exact in float32, any order) + bias[n]   (one float32 rounding, as numpy's)
#   dx[:, sel(n)] = the kernel's bf16 logit gradient of channel n, bit for bit (a sum of one non-zero product)
# Bounds, written before the first GPU run:
#   loss parts              1e-4 relative (TOL)
#   logit gradient          element-wise one bf16 rounding of the oracle's float32 value: |got - g| <= 2^-8 |g| + 2^-8 median|g|
#                           (helpers.assert_elem_close); an element of a cell matched by k >= 2 targets is summed in bf16 by k
#                           packed atomics in any order: + k 2^-8 max|g over that cell's channels|
#   bias gradient           float32 sums of the unrounded terms: 1e-5 of the level's max
#   weight gradient         dY^T X with the kernel's own bf16 dY (read back through dx) in float64, one bf16 rounding: 2^-8 |dw| + 2^-8 median|dw|

def _head_case(p_list, anchors_l, K_list, pieces=1, two_slice_level=None, bias_scale=0.0, seed=0):
    """p_list: per level float32 [B, na, H, W, no] logits-before-bias; with pieces == 1 they must be bf16-representable, with
    pieces == 3 any float32 (split into three bf16 pieces that sum back exactly).  Returns per level (x [B, K, H, W] bf16
    channels-last, w [N, K, 1, 1] bf16 selection weights, bias [N] float32, sel [pieces][N] input channel of each piece)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for l, (p, K) in enumerate(zip(p_list, K_list)):
        B, na, Hh, Ww, no = p.shape
        N = na * no
        rows = p.permute(0, 2, 3, 1, 4).reshape(B, Hh, Ww, N)           # channel n = a * no + c
        parts, rest = [], rows.clone()
        for _ in range(pieces):
            hi = rest.to(torch.bfloat16)      # round to nearest: the signed remainder loses >= 8 bits per piece
            parts.append(hi)
            rest = rest - hi.float()
        assert float(rest.abs().max()) == 0.0, "logits are not representable by the requested number of bf16 pieces"
        sel = []
        for q in range(pieces):
            base = q * ((N + 31) // 32 * 32)
            idx = torch.arange(N) + base
            if two_slice_level == l:                                      # odd channels come from the second K-slice of 256
                idx = idx + (torch.arange(N) % 2) * 256
            sel.append(idx)
        assert int(max(int(s.max()) for s in sel)) < K
        x = torch.zeros(B, Hh, Ww, K, dtype=torch.bfloat16)
        w = torch.zeros(N, K, dtype=torch.bfloat16)
        for q in range(pieces):
            x[..., sel[q]] = parts[q]
            w[torch.arange(N), sel[q]] = 1.0
        bias = (torch.randn(N, generator=g) * bias_scale).float()
        out.append((x.permute(0, 3, 1, 2), w.view(N, K, 1, 1), bias, sel))
    return out


def _run_default_head_path(gpu, ops, case, strides, anchors, slots, M, na, no, gains=(0.05, 1.0, 0.5), focal=None):
    xs, ws, bs, preds = [], [], [], []
    for x, w, b, _ in case:
        xd = x.to(gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd = w.to(gpu).requires_grad_(True)
        bd = b.to(gpu).requires_grad_(True)
        assert ops.head_conv_supported(xd, wd, bd)
        y = ops.head_conv(xd, wd, bd)
        B, P, Hh, Ww = y.shape
        assert P == ops.head_conv_pitch(na * no) and P % 8 == 0
        preds.append(y.as_strided((B, na, Hh, Ww, no), (Hh * Ww * P, no, Ww * P, P, 1)))
        xs.append(xd), ws.append(wd), bs.append(bd)
    total, parts = ops.yolo_loss(preds, strides, anchors, slots, M, *gains, biases=bs, heads=list(zip(xs, ws)), focal=focal)
    logits = [p.detach().cpu().numpy().copy() for p in preds]
    total.backward()
    ops.flush_wgrad()
    torch.cuda.synchronize()
    return parts.cpu().numpy(), logits, [x.grad for x in xs], [w.grad for w in ws], [b.grad for b in bs]


def _check_default_head_path(case, got, ref_loss, ref_grads, slot_cell, na, no, what):
    parts, _, dxs, dws, dbs = got
    assert H.max_rel_elem(parts, ref_loss, 1e-6) < TOL, (what, parts, ref_loss)
    for l, ((x, w, b, sel), dx, dw, db, g) in enumerate(zip(case, dxs, dws, dbs, ref_grads)):
        B, _, Hh, Ww, _ = g.shape
        N = na * no
        g_rows = np.ascontiguousarray(g.transpose(0, 2, 3, 1, 4)).reshape(B * Hh * Ww, N).astype(np.float64)
        dx_rows = dx.float().permute(0, 2, 3, 1).reshape(B * Hh * Ww, -1).cpu().numpy().astype(np.float64)
        # cells matched by k >= 2 targets: k bf16 atomic adds in any order
        sc = slot_cell[l]
        sc = sc[sc >= 0]
        n_match = np.bincount(sc, minlength=B * na * Hh * Ww).reshape(B, na, Hh, Ww)
        k_cell = np.ascontiguousarray(n_match.transpose(0, 2, 3, 1)).reshape(B * Hh * Ww, na)      # [row][anchor]
        cell_max = np.abs(g_rows.reshape(-1, na, no)).max(-1)
        extra = np.repeat(np.where(k_cell >= 2, k_cell, 0) * cell_max, no, axis=1) * 2.0 ** -8
        nz = np.abs(g_rows[g_rows != 0])
        floor = 2.0 ** -8 * float(np.median(nz))
        for q, idx in enumerate(sel):
            got_g = dx_rows[:, idx.numpy()]
            tol = 2.0 ** -8 * np.abs(g_rows) + floor + extra
            bad = np.abs(got_g - g_rows) > tol
            assert not bad.any(), (what, l, q, int(bad.sum()), float((np.abs(got_g - g_rows) / tol).max()))
        # input channels no output reads get no gradient
        unused = np.ones(dx_rows.shape[1], bool)
        for idx in sel:
            unused[idx.numpy()] = False
        assert float(np.abs(dx_rows[:, unused]).max() if unused.any() else 0.0) == 0.0
        ref_b = g_rows.sum(0)
        assert float(np.abs(db.cpu().numpy() - ref_b).max()) <= 1e-5 * float(np.abs(ref_b).max()), (what, l)
        # weight gradient from the kernel's OWN bf16 logit gradient (as read back) and x, float64
        dy_rows = dx_rows[:, sel[0].numpy()]
        x_rows = x.float().permute(0, 2, 3, 1).reshape(B * Hh * Ww, -1).numpy().astype(np.float64)
        H.assert_elem_close(dw.float().view(N, -1).cpu().numpy(), dy_rows.T @ x_rows, 2.0 ** -8, f"{what}: dw level {l}")
    return True


def _bf16_logits(B, size, na, no, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn((B, na, size // s, size // s, no), generator=g) * scale).to(torch.bfloat16).float() for s in H.YOLO_STRIDES]


@pytest.mark.parametrize("nc,K_list,two_slice", [(80, (256, 256, 512), 2), (1, (32, 64, 128), None), (20, (128, 128, 256), None)])
def test_default_training_head_path_against_the_oracle(gpu, ops, nc, K_list, two_slice):
    """hd_head_conv_fwd -> loss on padded rows -> bf16 gradient rows -> hd_head_conv_bwd against co.yolo_assign + co.yolo_loss on
    bit-identical float32 logits.  24 targets on a 128 px image: cells matched by three and more targets exist (asserted);
    nc = 80: pitch 256, the 4 x 4 level with Cin = 512 (two K-slices of the forward); nc = 1: no = 6 < 8 channels per 16-byte
    word (pitch 32); nc = 20: pitch 128."""
    B, size, na, M = 2, 128, 3, 24
    no = nc + 5
    p_list = _bf16_logits(B, size, na, no, 300 + nc)
    case = _head_case(p_list, H.YOLO_ANCHORS, K_list, 1, two_slice, bias_scale=0.5, seed=nc)
    gtb, gtc, mv = _yolo_targets(B, size, nc, 301 + nc, "coco", M)
    shapes = H.yolo_shapes(size, no)
    rc, rt, rl = co.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, gtb.numpy(), gtc.numpy(), mv.numpy(), 4.0)
    assert max(int(np.bincount(c[c >= 0]).max()) for c in rc if (c >= 0).any()) >= 3
    slots = ops.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, dev(gtb, gpu), dev(gtc, gpu), dev(mv, gpu), 4.0)
    got = _run_default_head_path(gpu, ops, case, H.YOLO_STRIDES, H.YOLO_ANCHORS, slots, M, na, no)
    # the oracle's inputs: the float32 logits as numpy forms them (selected bf16 value + bias, one rounding) — and they must be
    # the very bits hd_head_conv_fwd produced
    logits = [(p.numpy() + c[2].numpy().reshape(1, na, 1, 1, no)).astype(np.float32) for p, c in zip(p_list, case)]
    for a, b in zip(got[1], logits):
        assert np.array_equal(a, b)
    r_loss, r_grads, _, _ = co.yolo_loss(logits, H.YOLO_STRIDES, H.YOLO_ANCHORS, rc, rt, rl, M)
    _check_default_head_path(case, got, r_loss, r_grads, rc, na, no, f"nc={nc}")


def test_default_training_head_path_with_two_objectness_channels_in_one_word(gpu, ops):
    """nc = 1 and FOUR anchors per level: no = 6, so anchors 2 and 3 have their objectness channels 16 and 22 in the same 8-channel
    word of the bf16 gradient row (yolo_grad_dense_bf16_kernel must fold every anchor of a word; round 4's kernel kept the last
    match only and dropped the other's objectness and bias gradient)."""
    B, size, na, nc, M = 2, 128, 4, 1, 12
    no = nc + 5
    anchors = [[[10, 13], [16, 30], [33, 23], [20, 20]], [[30, 61], [62, 45], [59, 119], [50, 50]],
               [[116, 90], [156, 198], [373, 326], [100, 100]]]
    g = torch.Generator().manual_seed(77)
    p_list = [(torch.randn((B, na, size // s, size // s, no), generator=g)).to(torch.bfloat16).float() for s in H.YOLO_STRIDES]
    case = _head_case(p_list, anchors, (32, 32, 64), 1, None, bias_scale=0.5, seed=5)
    gtb, gtc, mv = _yolo_targets(B, size, nc, 78, "coco", M)
    shapes = [(na, size // s, size // s, no) for s in H.YOLO_STRIDES]
    rc, rt, rl = co.yolo_assign(shapes, H.YOLO_STRIDES, anchors, gtb.numpy(), gtc.numpy(), mv.numpy(), 4.0)
    slots = ops.yolo_assign(shapes, H.YOLO_STRIDES, anchors, dev(gtb, gpu), dev(gtc, gpu), dev(mv, gpu), 4.0)
    assert np.array_equal(slots[0].cpu().numpy(), rc)
    got = _run_default_head_path(gpu, ops, case, H.YOLO_STRIDES, anchors, slots, M, na, no)
    logits = [(p.numpy() + c[2].numpy().reshape(1, na, 1, 1, no)).astype(np.float32) for p, c in zip(p_list, case)]
    for a, b in zip(got[1], logits):
        assert np.array_equal(a, b)
    r_loss, r_grads, _, _ = co.yolo_loss(logits, H.YOLO_STRIDES, anchors, rc, rt, rl, M)
    _check_default_head_path(case, got, r_loss, r_grads, rc, na, no, "na=4 nc=1")
    # and specifically: every anchor's objectness gradient is there
    for l, g_ in enumerate(r_grads):
        dx = got[2][l].float().permute(0, 2, 3, 1).cpu().numpy()
        for a in range(na):
            ref = g_[:, a, :, :, 4]
            assert np.abs(ref).max() > 0 and np.abs(dx[..., a * no + 4] - ref).max() <= 2.0 ** -7 * np.abs(ref).max(), (l, a)


def test_golden_loss_case_through_the_default_training_head_path(gpu, ops):
    """tests/golden/hotpath_v1.npz's loss case (arbitrary float32 logits, nc = 4) through the same chain: every logit is split into
    three bf16 pieces that three selection weights add back exactly (Cin = 128), so hd_head_conv_fwd reproduces the fixture's
    logits bit for bit; loss against y_loss, bf16 gradient rows against y_g*."""
    gold = np.load(H.GOLDEN)
    p_list = [torch.from_numpy(gold[f"y_p{l}"]) for l in range(3)]
    B, na, _, _, no = p_list[0].shape
    case = _head_case(p_list, H.YOLO_ANCHORS, (128, 128, 128), 3, None, bias_scale=0.0)
    rc, rt, rl = gold["y_cell"], gold["y_tbox"], gold["y_cls"]
    M = gold["y_gtc"].shape[1]
    shapes = [(na, p.shape[2], p.shape[3], no) for p in p_list]
    slots = ops.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, torch.from_numpy(gold["y_gtb"]).to(gpu),
                            torch.from_numpy(gold["y_gtc"]).to(gpu), torch.from_numpy(gold["y_mv"]).to(gpu), 4.0)
    assert np.array_equal(slots[0].cpu().numpy(), rc)
    got = _run_default_head_path(gpu, ops, case, H.YOLO_STRIDES, H.YOLO_ANCHORS, slots, M, na, no)
    for a, l in zip(got[1], range(3)):
        assert np.array_equal(a, gold[f"y_p{l}"])
    # the fixture was made with the default gains and grad_scale 1 (tests/golden/make_golden.py); tests/test_oracle.py pins the
    # oracle to it on the CPU side
    _check_default_head_path(case, got, gold["y_loss"], [gold[f"y_g{l}"] for l in range(3)], rc, na, no, "golden")


def test_roi_align_backward_gather_takes_inverted_boxes_like_the_forward(gpu, ops):
    """aligned = 1 does not clamp a RoI's extent: x2 < x1 or y2 < y1 gives negative bins whose samples run down from the start
    corner.  The forward and the scatter backward include them; the gather's cell window must span them too (round 4's prep kernel
    assumed bins >= 0 and dropped such a RoI's gradient).  Gather == scatter (float32 maps, 2e-5) and == the oracle's backward
    (1e-4 element-wise) on a list with inverted boxes."""
    strides = [4, 8, 16, 32]
    B, C, size, R = 2, 64, 256, 40
    feats, rois = _roi_case(913, B, C, size, R, strides)
    inv = rois.clone()
    inv[::3, 1], inv[::3, 3] = rois[::3, 3], rois[::3, 1]          # x inverted
    inv[1::4, 2], inv[1::4, 4] = rois[1::4, 4], rois[1::4, 2]      # y inverted
    scales = [1.0 / s for s in strides]
    res = {}
    go = None
    for mode in ("gather", "scatter"):
        ops.RoiBwdSwitch.gather = mode == "gather"
        try:
            fd = [dev(f, gpu).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
            out = ops.roi_align(fd, dev(inv, gpu), 7, scales, 2, True, 2, 224.0 / 4, 4, channels_last_out=True)
            if go is None:
                go = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).to(gpu).contiguous(memory_format=torch.channels_last)
            out.backward(go)
            res[mode] = [f.grad.clone() for f in fd]
        finally:
            ops.RoiBwdSwitch.gather = True
    ref = co.roi_align_bwd([tuple(f.shape) for f in feats], inv.numpy(), np.ascontiguousarray(go.cpu().numpy()), scales, 2, True, 2, 224.0 / 4, 4)
    assert any(float(np.abs(r).max()) > 0 for r in ref)
    for a, c, r in zip(res["gather"], res["scatter"], ref):
        err = (a - c).abs()
        assert bool((err <= 2e-5 * c.abs() + 2e-5 * c.abs().mean()).all()), float(err.max())
        H.assert_elem_close(a.cpu().numpy(), r, TOL, "gather backward with inverted boxes")
