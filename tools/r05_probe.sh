#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_pointwise_conv.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5/pw_tests.log 2>&1 || { tail -30 gpurun_out/r5/pw_tests.log; exit 1; }
tail -2 gpurun_out/r5/pw_tests.log
timeout -k 10 300 python tools/pw_long_k_probe.py > gpurun_out/r5/pw_long_k.log 2>&1
echo rc=$?; grep -v amdgpu.ids gpurun_out/r5/pw_long_k.log | tail -20
