set -x
mkdir -p gpurun_out/r5
./tools/micro/bin/buffer_soffset_probe > gpurun_out/r5/soffset_probe.log 2>&1; cat gpurun_out/r5/soffset_probe.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "default_training or golden_loss_case or inverted or two_objectness" > gpurun_out/r5/new_parity.log 2>&1; tail -15 gpurun_out/r5/new_parity.log
