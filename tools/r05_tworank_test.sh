mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_dist_graph.py::test_bench_two_rank_rehearsal_prints_one_compact_line_and_exits -k "bias_act or cat_inplace or cli or conv3x3 or dist_graph" > gpurun_out/r5/tworank_suite_prefix.log 2>&1
grep "per-bucket\|passed\|failed" gpurun_out/r5/tworank_suite_prefix.log | cut -c1-600
