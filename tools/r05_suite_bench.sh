mkdir -p gpurun_out/r5
timeout -k 10 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5/gpu_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r5/gpu_tests.log
if grep -q "Memory access fault" gpurun_out/r5/gpu_tests.log; then echo FAULT; exit 1; fi
[ $rc -ne 0 ] && exit $rc
( time python bench.py > gpurun_out/r5/bench_default.log 2> gpurun_out/r5/bench_default.err ) 2> gpurun_out/r5/bench_default.time
rc=$?
tail -c 1500 gpurun_out/r5/bench_default.log; cat gpurun_out/r5/bench_default.time
cp bench_detail.json gpurun_out/r5/bench_detail.json 2>/dev/null
exit $rc
