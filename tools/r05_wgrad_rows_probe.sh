#!/bin/bash
# row weight-gradient kernels: exact tests, then the per-shape probe (row kernels only, and the automatic choice)
mkdir -p gpurun_out/r5b
timeout -k 10 400 python -m pytest tests/test_gpu_conv3x3_wgrad.py tests/test_gpu_conv3x3s2_dgrad.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5b/wgrad_tests2.log 2>&1 || { tail -40 gpurun_out/r5b/wgrad_tests2.log; exit 1; }
tail -2 gpurun_out/r5b/wgrad_tests2.log
WGRAD3_SELECT=1 timeout -k 10 200 python tools/wgrad3_probe.py > gpurun_out/r5b/wgrad_probe_rows2.log 2>&1 || { tail -20 gpurun_out/r5b/wgrad_probe_rows2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5b/wgrad_probe_rows2.log | tail -10
