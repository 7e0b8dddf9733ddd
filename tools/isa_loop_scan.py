"""Static check of the hot loops: compiles csrc/*.hip for gfx950 with -save-temps and, for every kernel that contains MFMAs, lists
its innermost MFMA loops with the number of VMEM loads and barriers inside and the `s_waitcnt vmcnt(N)` sequence the compiler
placed there.  A loop whose loads are followed by `vmcnt(0)` waits while younger loads could still be in flight is a loop whose
lookahead the compiler has cancelled — the usual causes found in this tree: loads under a lane predicate or inside a runtime
condition (the compiler cannot order them against the other register set: csrc/pointwise_conv.hip, conv3x3_wgrad_kernel), and a
second copy of the same loads outside the loop (register copies + a full wait at the loop head).  No GPU needed.

    python tools/isa_loop_scan.py [name fragment ...]      e.g.  python tools/isa_loop_scan.py conv3x3_wgrad pointwise_bwd
"""
import concurrent.futures as cf
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "heltondetection_amd", "csrc")


def compile_all(out):
    def one(src):
        base = os.path.basename(src)[:-4]
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                        f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-c", src, "-o", os.path.join(out, base + ".o"),
                        "--cuda-device-only", "-save-temps=obj"], capture_output=True, check=True)
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(one, sorted(glob.glob(os.path.join(CSRC, "*.hip")))))


def scan(path, want):
    S = open(path).read().split("\n")
    for st in [i for i, l in enumerate(S) if re.match(r"^_Z\w+:", l)]:
        end = st
        while end < len(S) and not S[end].startswith(".Lfunc_end"):
            end += 1
        body = S[st:end]
        if not any("v_mfma" in l for l in body):
            continue
        name = subprocess.run(["c++filt", S[st].split(":")[0]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r">\(.*", ">", name.replace("(anonymous namespace)::", "")).replace("void ", "", 1)
        if want and not any(w in name for w in want):
            continue
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))

        def has(a, b):
            return any("v_mfma" in l for l in body[a:b + 1])

        print(name[:120])
        for a, b in loops:
            if not has(a, b) or any(a2 >= a and b2 <= b and (a2, b2) != (a, b) and has(a2, b2) for a2, b2 in loops):
                continue
            seg = body[a:b + 1]
            waits = [re.search(r"vmcnt\((\d+)\)", l).group(1) for l in seg if "vmcnt" in l]
            n_load = sum(bool(re.search(r"(buffer|global)_load", l)) for l in seg)
            print(f"     loop: {sum('v_mfma' in l for l in seg)} mfma, {n_load} vmem loads, {sum('s_barrier' in l for l in seg)} barriers, "
                  f"vmcnt waits: {' '.join(waits) if waits else '-'}")


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        compile_all(d)
        for f in sorted(glob.glob(os.path.join(d, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
            scan(f, sys.argv[1:])
