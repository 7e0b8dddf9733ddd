mkdir -p gpurun_out/r5
bash tools/r05_conv3.sh > gpurun_out/r5/conv3_run.log 2>&1
grep -B30 "kernel v1" gpurun_out/r5/conv3_run.log | grep "passed\|failed\|fwd\|dgrad"
python tools/conv3_stamps.py > gpurun_out/r5/conv3_stamps_run.log 2>&1
grep "stats=\|chunk 2\|phase dur" gpurun_out/r5/conv3_stamps_run.log
