#!/bin/bash
# row weight-gradient kernels: exact tests, the per-shape probe (automatic choice; config 3's shapes), two quick headline runs
mkdir -p gpurun_out/r5b
timeout -k 10 400 python -m pytest tests/test_gpu_conv3x3_wgrad.py tests/test_gpu_conv3x3s2_dgrad.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5b/wgrad_tests3.log 2>&1 || { tail -40 gpurun_out/r5b/wgrad_tests3.log; exit 1; }
tail -2 gpurun_out/r5b/wgrad_tests3.log
timeout -k 10 200 python tools/wgrad3_probe.py > gpurun_out/r5b/wgrad_probe_auto3.log 2>&1 || { tail -20 gpurun_out/r5b/wgrad_probe_auto3.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5b/wgrad_probe_auto3.log | tail -10
WGRAD3_SHAPES=frcnn WGRAD3_SELECT=1 timeout -k 10 200 python tools/wgrad3_probe.py 2>&1 | grep " s[12] " > gpurun_out/r5b/wgrad_probe_frcnn3.log; cat gpurun_out/r5b/wgrad_probe_frcnn3.log
B="python bench.py --steps 40 --warmup 10 --no-legs --no-micro --no-cpu-baseline --no-roofline"
for rep in 1 2 3; do timeout -k 10 200 $B 2>&1 | grep "timed region" || exit 1; done | tee gpurun_out/r5b/quick_steps3.log
