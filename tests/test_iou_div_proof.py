"""The IoU kernels replace the IEEE division inter / union by v_rcp_f32 + six FMA-class operations
(csrc/hd_common.h: hd_quot_fast*).  NMS keep masks, argmax indices and the materialised IoU tensor are compared
bit-for-bit with the oracle, which divides — so the replacement must return the same bits, for every reciprocal
estimate the hardware may produce (1 ulp).  tests/proofs/iou_div_proof.c checks that on the CPU: exhaustively
over all 2^23 denominator significands for the reciprocal step, exhaustively over all numerators for the few
exception denominators, and over 2^23 x 101 structured pairs in both roles (see its header for the argument).
The GPU side of the same claim is tests/test_gpu_parity.py (bit-exact IoU / assignment / NMS vs the oracle)."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "proofs", "iou_div_proof.c")


def _cpu_has_fma() -> bool:
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


@pytest.mark.timeout(900)
def test_division_free_quotient_is_bit_exact(tmp_path):
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = str(tmp_path / "iou_div_proof")
    flags = ["-O2", "-fopenmp"] + (["-mfma"] if _cpu_has_fma() else [])
    r = subprocess.run([gcc, *flags, SRC, "-o", exe, "-lm"], capture_output=True, text=True)
    if r.returncode != 0:  # no OpenMP runtime: single-threaded build
        r = subprocess.run([gcc, *[f for f in flags if f != "-fopenmp"], SRC, "-o", exe, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, OMP_NUM_THREADS=str(min(8, os.cpu_count() or 1)))
    out = subprocess.run([exe], capture_output=True, text=True, env=env)
    print(out.stdout)
    assert out.returncode == 0 and "PROOF OK" in out.stdout, out.stdout + out.stderr
    # the only denominators that may miss are the all-ones significands the kernels send to the IEEE division
    assert not re.search(r"UNEXPECTED", out.stdout)
    m = re.search(r"exception set X\):((?: [0-9a-f]{6})*)", out.stdout)
    assert m and "7fffff" in m.group(1).split()


def test_kernel_sequence_matches_the_proved_one():
    """Lockstep check: the operations of hd_quot_fast in csrc/hd_common.h are the ones the C file proves."""
    hdr = open(os.path.join(HERE, "..", "heltondetection_amd", "csrc", "hd_common.h")).read()
    body = hdr[hdr.index("__device__ __forceinline__ float hd_quot_fast("):]
    body = body[body.index("{") + 1: body.index("}")]
    steps = [s.strip() for s in body.split(";") if "=" in s or "return" in s]
    want = ["y0 = __builtin_amdgcn_rcpf(u)", "e = fmaf(-u, y0, 1.0f)", "t = fmaf(e, e, e)", "y1 = fmaf(t, y0, y0)",
            "q0 = n * y1", "r = fmaf(-u, q0, n)", "return fmaf(r, y1, q0)"]
    got = [re.sub(r"^const float ", "", s) for s in steps]
    assert got == want, got
    c = open(SRC).read()
    for frag in ("fmaf(-u, y0, 1.0f)", "fmaf(e, e, e)", "fmaf(t, y0, y0)", "n * y1", "fmaf(-u, q0, n)", "fmaf(r, y1, q0)"):
        assert frag in c
