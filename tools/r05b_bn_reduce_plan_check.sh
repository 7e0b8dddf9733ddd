#!/bin/bash
# bn_bwd_reduce at 4 vectors per lane: the tests that run BatchNorm backward, then two quick headline runs
mkdir -p gpurun_out/r5b
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syncbn.py tests/test_gpu_cat_inplace.py tests/test_gpu_pointwise_conv.py tests/test_gpu_pointwise_bwd.py tests/test_gpu_grad_fidelity.py -x -q -m gpu -p no:cacheprovider -k "bn or BN or batch or norm or cat or node or fused or fidelity or layer" > gpurun_out/r5b/bn_tests2.log 2>&1 || { tail -30 gpurun_out/r5b/bn_tests2.log; exit 1; }
tail -1 gpurun_out/r5b/bn_tests2.log
B="python bench.py --steps 40 --warmup 10 --no-legs --no-micro --no-cpu-baseline --no-roofline"
for rep in 1 2; do timeout -k 10 200 $B 2>&1 | grep "timed region" || exit 1; done | tee gpurun_out/r5b/quick_steps4.log
