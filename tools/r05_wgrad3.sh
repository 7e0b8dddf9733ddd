mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_conv3x3_wgrad.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5/wgrad3_tests.log 2>&1 || { tail -30 gpurun_out/r5/wgrad3_tests.log; exit 1; }
tail -1 gpurun_out/r5/wgrad3_tests.log
python tools/wgrad3_probe.py 2>&1 | grep -v amdgpu.ids
HD_WGRAD3_WGS=256 python tools/wgrad3_probe.py 2>&1 | grep -v amdgpu.ids
HD_WGRAD3_V1=1 python tools/wgrad3_probe.py 2>&1 | grep -v amdgpu.ids
