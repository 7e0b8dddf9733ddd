mkdir -p gpurun_out/r5
for sel in 0 1 2; do WGRAD3_SELECT=$sel python tools/wgrad3_probe.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r5/wgrad3_probe.log
cat gpurun_out/r5/wgrad3_probe.log
