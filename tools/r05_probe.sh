#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_gpu_stem_conv.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r5/stem_tests.log 2>&1 || { tail -40 gpurun_out/r5/stem_tests.log; exit 1; }
tail -2 gpurun_out/r5/stem_tests.log
bash tools/r05_quick_bench.sh
grep stem gpurun_out/r5/layer_detail.jsonl
