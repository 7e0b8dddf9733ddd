"""CPU-side guard of the hot kernels' register budgets: hipcc's kernel-resource-usage remarks for the convolution sources must show
no scratch (spilled registers) in any instantiation the train step launches, and the occupancy each kernel was laid out for.

Round 4's review found 12 - 28 bytes / lane of scratch in two 3x3 weight-gradient instantiations; round 5 removed them (buffer
row loads: one offset register per tensor instead of a 64-bit address per load).  A later edit that pushes one of these kernels
over its budget fails here, without a GPU."""
import concurrent.futures as cf
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "heltondetection_amd", "csrc")

# source -> [(regex on the demangled kernel name, minimum waves per SIMD the kernel is laid out for)]
HOT = {
    "pointwise_conv.hip": [(r"^conv3x3_wgrad_kernel<(20|26|40|52|80)>", 2), (r"^conv3x3_wgrad_kernel<104>", 1),
                           (r"^conv3x3s2_wgrad_kernel<(20|26|40)>", 2), (r"^conv3x3s2_wgrad_kernel<(52|80|104)>", 1),
                           (r"^pointwise_stream_kernel<", 1), (r"^pointwise_wgrad_kernel<", 2)],
    "pointwise_bwd.hip": [(r"^pointwise_bwd_kernel<", 1)],
    "conv3x3.hip": [(r"^conv3x3_tap_kernel<", 1)],
}


def _remarks(src):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", f"-I{os.path.join(ROOT, 'include')}",
                          f"-I{CSRC}", "-c", os.path.join(CSRC, src), "-o", os.devnull, "--cuda-device-only",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        for key, pat in (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.splitlines()
    clean = lambda n: re.sub(r">\(.*", ">", n.replace("(anonymous namespace)::", "")).replace("void ", "", 1)
    return {clean(n): v for n, v in zip(names, kernels.values())}


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_hot_convolution_kernels_are_scratch_free_at_their_occupancy():
    with cf.ThreadPoolExecutor(max_workers=len(HOT)) as ex:
        reports = dict(zip(HOT, ex.map(_remarks, HOT)))
    problems, seen = [], 0
    for src, rules in HOT.items():
        for pat, min_occ in rules:
            hits = {n: v for n, v in reports[src].items() if re.search(pat, n)}
            assert hits, f"{src}: no kernel matches {pat} (renamed? update HOT)"
            for n, v in hits.items():
                seen += 1
                if v.get("scratch", 0) != 0:
                    problems.append(f"{n}: {v['scratch']} bytes / lane of scratch ({v.get('vgprs')} VGPRs + {v.get('agprs')} AGPRs)")
                if v.get("occupancy", 0) < min_occ:
                    problems.append(f"{n}: occupancy {v.get('occupancy')} waves / SIMD, laid out for >= {min_occ}")
    assert seen >= 40, seen
    assert not problems, "\n".join(problems)
