"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/hd_ops.h
declares (and nothing in the binding is missing from the header), and the product has no CPU path."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "hd_ops.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(hd_[a-z0-9_]+)\s*\(", src))


def _header_abi_digest():
    """sha256 over everything a caller compiles against: the header without comments and blank runs (prototypes, structs,
    constants).  Any change of it is an ABI change."""
    import hashlib
    src = open(os.path.join(ROOT, "include", "hd_ops.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"\s+", " ", src).strip()
    return hashlib.sha256(src.encode()).hexdigest()[:16]


# version -> digest of the header it was cut from.  A header edit without a new version (hd_abi_version in csrc/lib.hip,
# ABI_VERSION in _lib.py, a line here) fails this test: a stale libhd_ops.so must fail with "rebuild", not AttributeError
# (round 4 added 17 entry points at version 29).
ABI_HEADER_DIGESTS = {30: "1f8ec122881dfc5b", 31: "7244275d8df1b40b", 32: "5edfa93a8d0aff03", 33: "f9939a4b68f2ccd1", 34: "f6ca61cea8bf4baf"}


def test_abi_version_is_bumped_with_the_header():
    from heltondetection_amd import _lib
    digest = _header_abi_digest()
    assert _lib.ABI_VERSION in ABI_HEADER_DIGESTS, f"ABI_VERSION {_lib.ABI_VERSION} has no digest line in tests/test_abi.py"
    assert ABI_HEADER_DIGESTS[_lib.ABI_VERSION] == digest, (
        f"include/hd_ops.h changed (digest {digest}) but the ABI version is still {_lib.ABI_VERSION}: bump hd_abi_version / "
        "ABI_VERSION and add the new (version, digest) line")
    src = open(os.path.join(ROOT, "heltondetection_amd", "csrc", "lib.hip")).read()
    assert re.search(r"hd_abi_version\(void\)\s*\{\s*return\s+%d\s*;" % _lib.ABI_VERSION, src)


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from heltondetection_amd import _lib
    lib = _lib.load()
    declared = _header_functions()
    assert declared, "no functions parsed from the header"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in hd_ops.h but not exported"
    assert lib.hd_abi_version() == _lib.ABI_VERSION
    assert lib.hd_status_str(0) == b"ok"
    assert lib.hd_status_str(-4).startswith(b"workspace")


def test_workspace_queries_are_pure_host_arithmetic():
    from heltondetection_amd import _lib
    lib = _lib.load()
    assert lib.hd_iou_assign_workspace_bytes(2, 100, 7) == 2 * 7 * 8
    assert lib.hd_nms_workspace_bytes(0, 10, 0) == 0
    a = lib.hd_nms_workspace_bytes(2, 1000, 0)
    b = lib.hd_nms_workspace_bytes(4, 1000, 0)
    assert 0 < a < b
    assert lib.hd_nms_workspace_bytes(2, 1000, 100) < a  # clamped candidates shrink the mask
    assert lib.hd_wbf_workspace_bytes(3, 50) > 0
    assert lib.hd_rpn_proposals_workspace_bytes(2, 5000, 5, 1000, 300) > 0
    assert lib.hd_yolo_nms_workspace_bytes(2, 25200, 30000, 300) > 0


def test_host_helper_anchor_base_matches_oracle():
    import numpy as np
    from heltondetection_amd import ops
    from oracle import c_oracle as co
    got = ops.anchor_base([32.0, 64.0], [0.5, 1.0, 2.0]).numpy()
    assert np.array_equal(got, co.anchor_base([32.0, 64.0], [0.5, 1.0, 2.0]))


def test_argument_validation_without_a_gpu():
    """Status codes are produced on the host before any launch (no GPU needed)."""
    import ctypes as C
    from heltondetection_amd import _lib
    lib = _lib.load()
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    assert lib.hd_pairwise_iou(None, 0, None, None, 1, 4, 4, None, None) == -1       # HD_ERR_NULL
    assert lib.hd_pairwise_iou(p, 0, p, None, -1, 4, 4, p, None) == -2               # HD_ERR_SHAPE
    assert lib.hd_pairwise_iou(C.c_void_p(p.value + 4), 0, p, None, 1, 4, 4, p, None) == -3  # HD_ERR_ALIGN
    assert lib.hd_iou_assign(p, 0, p, None, 1, 4, 4, p, p, p, p, None, 0, None) == -4  # HD_ERR_WORKSPACE
    assert lib.hd_nms_batched(p, p, None, None, 1, 4, 0.0, 0.5, 0, 4, p, p, None, 0, None) == -4
    assert lib.hd_pairwise_iou(None, 0, None, None, 0, 0, 0, None, None) == 0        # empty problem is OK


def test_product_has_no_cpu_path():
    from heltondetection_amd import _lib, ops
    a = torch.zeros((1, 4, 4))
    with pytest.raises(_lib.HdError):
        ops.pairwise_iou(a, a)
    with pytest.raises(_lib.HdError):
        ops.nms_batched(a, torch.zeros((1, 4)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "heltondetection_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in txt and "import oracle" not in txt and "hd_oracle" not in txt.replace(
                    "oracle/hd_oracle.c", ""), f"{f} references the oracle"


def test_detect_convolution_host_arithmetic_and_validation():
    """hd_head_conv_*: the padded pitch, the support predicates and the argument checks are host code (no GPU touched)."""
    from heltondetection_amd import _lib
    lib = _lib.load()
    assert [lib.hd_head_conv_pitch(c) for c in (1, 18, 32, 33, 75, 128, 129, 255, 256)] == [32, 32, 32, 64, 128, 128, 256, 256, 256]
    assert lib.hd_head_conv_pitch(257) == 0 and lib.hd_head_conv_pitch(0) == 0
    assert lib.hd_head_conv_fwd_supported(102400, 128, 255) == 1
    assert lib.hd_head_conv_fwd_supported(6400, 512, 255) == 1          # two K-slices of 256
    assert lib.hd_head_conv_fwd_supported(6400, 192, 255) == 0          # no resident-weight form: the library runs the forward
    assert lib.hd_head_conv_bwd_supported(6400, 192, 255) == 1
    assert lib.hd_head_conv_bwd_supported(6400, 128, 300) == 0
    assert lib.hd_head_conv_bwd_workspace_bytes(102400, 128, 255) == lib.hd_pointwise_bwd_workspace_bytes(102400, 128, 256) > 0
    assert lib.hd_head_conv_fwd(None, 0, None, None, 64, 128, 255, None, 0, None) == -1          # HD_ERR_NULL


def test_first_layer_and_long_reduction_entry_points_host_arithmetic_and_validation():
    """Round 5's late entry points: support predicates, the workspace query and the argument checks are host code (no GPU touched)."""
    import ctypes as C
    from heltondetection_amd import _lib
    lib = _lib.load()
    for fn in (lib.hd_conv3x3s2_fwd_supported, lib.hd_conv3x3s2_dgrad_supported, lib.hd_conv3x3s2_wgrad_c32_supported):
        assert fn(16, 160, 160, 32, 64) == 1
        assert fn(16, 208, 208, 32, 64) == 0            # 832^2: the library keeps the layer
        assert fn(16, 160, 160, 64, 128) == 0
        assert fn(0, 160, 160, 32, 64) == 0
    # one float32 slab of 64 x 9 x 32 per workgroup, ~256 workgroups of whole rows of one image
    assert lib.hd_conv3x3s2_wgrad_c32_workspace_bytes(16, 160, 160, 32, 64) == 16 * 16 * 64 * 9 * 32 * 4
    assert lib.hd_conv3x3s2_wgrad_c32_workspace_bytes(1, 3, 160, 32, 64) == 3 * 64 * 9 * 32 * 4
    assert lib.hd_conv3x3s2_wgrad_c32_workspace_bytes(16, 80, 80, 64, 128) == 0
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    assert lib.hd_conv3x3s2_fwd(None, 0, p, 1, 4, 160, 32, 64, p, 0, None, None) == -1                      # HD_ERR_NULL
    assert lib.hd_conv3x3s2_fwd(p, 0, p, 1, 4, 80, 32, 64, p, 0, None, None) == -2                          # HD_ERR_SHAPE
    assert lib.hd_conv3x3s2_fwd(p, 36, p, 1, 4, 160, 32, 64, p, 0, None, None) == -2                        # row stride not a multiple of 8
    assert lib.hd_conv3x3s2_dgrad(C.c_void_p(p.value + 2), 0, p, 1, 4, 160, 32, 64, p, 0, None) == -3       # HD_ERR_ALIGN
    assert lib.hd_conv3x3s2_wgrad_c32(p, 0, p, 0, 1, 4, 160, 32, 64, p, p, 16, None) == -4                  # HD_ERR_WORKSPACE
    # the 32 -> 32 layer at 160 pixels rides behind hd_conv3x3_fwd; the tap kernel's own shapes are unchanged
    assert lib.hd_conv3x3_fwd_supported(16, 160, 160, 32, 32) == 1
    assert lib.hd_conv3x3_fwd_supported(16, 160, 160, 32, 64) == 0 and lib.hd_conv3x3_fwd_supported(16, 80, 80, 32, 32) == 0
    assert lib.hd_conv3x3_fwd_supported(16, 80, 80, 64, 64) == 1
    # long reductions: Cin = 512 on at most 16 384 rows (pointwise_longk.hip) counts as a resident-weight form
    assert lib.hd_pointwise_conv_resident(6400, 512, 256) == 1 and lib.hd_pointwise_conv_resident(6400, 512, 512) == 1
    assert lib.hd_pointwise_conv_resident(25600, 512, 256) == 0 and lib.hd_pointwise_conv_resident(6400, 1024, 512) == 0
    assert lib.hd_pointwise_conv_resident(6400, 512, 192) == 0          # not a multiple of the 128-wide column tile
    # the flip launch: tap counts other than 1 and 9 are refused before anything is launched
    one = (C.c_void_p * 1)(p.value)
    n = (C.c_int * 1)(32)
    bad = (C.c_int * 1)(4)
    assert lib.hd_conv_flip_weights(1, one, one, n, n, bad, None) == -2
    assert lib.hd_conv_flip_weights(0, None, None, None, None, None, None) == 0
    assert lib.hd_conv_flip_weights(33, one, one, n, n, None, None) == -2
