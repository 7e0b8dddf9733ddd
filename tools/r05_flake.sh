#!/bin/bash
mkdir -p gpurun_out/r5
for i in 1 2 3; do
timeout -k 10 600 python -m pytest tests/test_gpu_dist_graph.py -x -q -s -m gpu -p no:cacheprovider -k "two_rank_overlapped" > gpurun_out/r5/tworank_$i.log 2>&1 || { echo "FAILED run $i"; }
grep -a "pre-train:\|per-bucket\|per-parameter\|passed\|failed" gpurun_out/r5/tworank_$i.log | cut -c1-600
done
