#!/bin/bash
# row weight-gradient kernels after the buffer-load / one-round-trip start change: their exact tests, the per-shape probe
# (row kernels only, then the automatic choice), then the full GPU suite and the default bench
mkdir -p gpurun_out/r5b
timeout -k 10 400 python -m pytest tests/test_gpu_conv3x3_wgrad.py tests/test_gpu_conv3x3s2_dgrad.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5b/wgrad_tests.log 2>&1 || { tail -40 gpurun_out/r5b/wgrad_tests.log; exit 1; }
tail -2 gpurun_out/r5b/wgrad_tests.log
WGRAD3_SELECT=1 timeout -k 10 200 python tools/wgrad3_probe.py > gpurun_out/r5b/wgrad_probe_rows.log 2>&1 || { tail -20 gpurun_out/r5b/wgrad_probe_rows.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5b/wgrad_probe_rows.log | tail -12
timeout -k 10 700 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5b/gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/r5b/gpu_tests.log
if grep -q "Memory access fault" gpurun_out/r5b/gpu_tests.log; then echo FAULT; exit 1; fi
[ $rc -ne 0 ] && exit $rc
( time python bench.py > gpurun_out/r5b/bench_default.log 2> gpurun_out/r5b/bench_default.err ) 2> gpurun_out/r5b/bench_default.time
rc=$?
tail -c 1200 gpurun_out/r5b/bench_default.log; cat gpurun_out/r5b/bench_default.time
cp bench_detail.json gpurun_out/r5b/bench_detail.json 2>/dev/null
exit $rc
