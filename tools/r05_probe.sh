#!/bin/bash
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests/test_gpu_dist_graph.py tests/test_gpu_models.py -x -q -m gpu -p no:cacheprovider -k "two_rank_overlapped or one_graph_replayed_optimizer_step" > gpurun_out/r5/flaky_tests.log 2>&1 || { tail -40 gpurun_out/r5/flaky_tests.log; exit 1; }
tail -2 gpurun_out/r5/flaky_tests.log
