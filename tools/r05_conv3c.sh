mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_conv3x3_wgrad.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5/conv3_tests.log 2>&1 || { tail -30 gpurun_out/r5/conv3_tests.log; exit 1; }
tail -1 gpurun_out/r5/conv3_tests.log
python tools/conv3_probe.py 2>&1 | grep "fwd\|dgrad"
for a in 1 2; do echo "HD_CONV3_ALT=$a:"; HD_CONV3_ALT=$a timeout -k 10 300 python -m pytest tests/test_gpu_conv3x3_wgrad.py -x -q -m gpu -p no:cacheprovider -k forward 2>&1 | tail -1; HD_CONV3_ALT=$a python tools/conv3_probe.py 2>&1 | grep "fwd\|dgrad"; done
