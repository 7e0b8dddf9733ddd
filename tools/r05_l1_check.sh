#!/bin/bash
# the first-block kernels: their tests, then tools/l1_probe.py
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_conv3x3s2_dgrad.py tests/test_gpu_conv3x3_wgrad.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5/l1_tests.log 2>&1 || { tail -40 gpurun_out/r5/l1_tests.log; exit 1; }
tail -2 gpurun_out/r5/l1_tests.log
timeout -k 10 300 python tools/l1_probe.py > gpurun_out/r5/l1_probe.log 2>&1
echo rc=$?; grep -v amdgpu.ids gpurun_out/r5/l1_probe.log | tail -12
