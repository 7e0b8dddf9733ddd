"""Builds libhd_ops.so (HIP, gfx950 only) in-tree: one hipcc -c per .hip, then one link.

The .so stays next to the sources (git-ignored, shipped to the GPU box by gpurun).  hipcc
cross-compiles without a GPU.  Flags that matter:
  --offload-arch=gfx950   the only target (no fat binary, no dual path)
  -ffp-contract=off       box arithmetic must round like oracle/hd_oracle.c
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libhd_ops.so")
OBJ_DIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-function", f"-I{os.path.join(ROOT, 'include')}", f"-I{HERE}"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _compile(src: str) -> str:
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    deps = [src] + glob.glob(os.path.join(HERE, "*.h")) + [os.path.join(ROOT, "include", "hd_ops.h"), __file__]
    if os.path.exists(obj) and os.path.getmtime(obj) >= _newest(deps):
        return obj
    cmd = ["hipcc", *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int = 8) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    if not srcs:
        raise RuntimeError("no .hip sources found")
    if force:
        for f in glob.glob(os.path.join(OBJ_DIR, "*.o")):
            os.remove(f)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < _newest(objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
