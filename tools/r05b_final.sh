#!/bin/bash
# last GPU call of round 5's second session: kernel stats + step trace of the final build, then the full GPU suite, smoke and
# the default bench.  Output under gpurun_out/r5b/ (summaries copied to profiles/r05b_*).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5b
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-legs --no-cpu-baseline"
rm -rf /tmp/ps1 /tmp/pt1
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps1 -- python3 bench.py $ARGS > gpurun_out/r5b/bench_profiled.log 2> gpurun_out/r5b/bench_profiled.err) || { echo "stats run failed"; tail -5 $R/gpurun_out/r5b/bench_profiled.err; exit 1; }
cp $(ls /tmp/ps1/*/*kernel_stats.csv | head -1) $R/gpurun_out/r5b/bench_command_kernel_stats.csv
echo "stats done"
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt1 -- python3 bench.py --steps 30 --warmup 8 --no-micro --no-cpu-baseline --no-roofline --no-legs > gpurun_out/r5b/trace_bench.log 2>&1) || { echo "trace run failed"; exit 1; }
python3 $R/tools/step_trace_parse_r02.py /tmp/pt1 200 > $R/gpurun_out/r5b/step_trace.txt 2>&1
head -4 $R/gpurun_out/r5b/step_trace.txt | cut -c1-150
rm -rf /tmp/ps1 /tmp/pt1
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r5b/gpu_tests_final.log 2>&1
rc=$?
tail -3 gpurun_out/r5b/gpu_tests_final.log
if grep -q "Memory access fault" gpurun_out/r5b/gpu_tests_final.log; then echo FAULT; exit 1; fi
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5b/smoke.log 2>&1 || { tail -5 gpurun_out/r5b/smoke.log; exit 1; }
tail -1 gpurun_out/r5b/smoke.log
( time python bench.py > gpurun_out/r5b/bench_default_final.log 2> gpurun_out/r5b/bench_default_final.err ) 2> gpurun_out/r5b/bench_default_final.time
rc=$?
tail -c 1200 gpurun_out/r5b/bench_default_final.log; cat gpurun_out/r5b/bench_default_final.time
cp bench_detail.json gpurun_out/r5b/bench_detail_final.json 2>/dev/null
exit $rc
