mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_conv3x3_wgrad.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5/conv3_tests.log 2>&1 || { tail -30 gpurun_out/r5/conv3_tests.log; exit 1; }
tail -3 gpurun_out/r5/conv3_tests.log
python tools/conv3_probe.py > gpurun_out/r5/conv3_probe_v2.log 2>&1
cat gpurun_out/r5/conv3_probe_v2.log
