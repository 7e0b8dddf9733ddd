# quick look at the step: the conv / model / fidelity tests, then the headline leg only
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_conv3x3_wgrad.py tests/test_gpu_grad_fidelity.py tests/test_gpu_models.py -x -q -m gpu -p no:cacheprovider -k "not faster_rcnn and not config5 and not yolov5l" > gpurun_out/r5/quick_tests.log 2>&1 || { tail -30 gpurun_out/r5/quick_tests.log; exit 1; }
tail -1 gpurun_out/r5/quick_tests.log
python bench.py --steps 30 --warmup 10 --no-legs --no-micro --no-cpu-baseline --layer-detail gpurun_out/r5/layer_detail.jsonl > gpurun_out/r5/bench_quick.log 2> gpurun_out/r5/bench_quick.err
tail -c 1200 gpurun_out/r5/bench_quick.log
grep "timed region" gpurun_out/r5/bench_quick.err
