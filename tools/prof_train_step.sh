cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf
# warm the MIOpen find db first (un-profiled), so the profiled run's first step is short
python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 6 --no-micro --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_warm.log 2>&1
echo "warm done" >> $GRAFT_REPO_ROOT/gpurun_out/prof_warm.log
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-micro --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_final_bench.log 2>&1
f=$(ls /tmp/pf/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_final_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/summarize_stats.py $f 48 30 > $GRAFT_REPO_ROOT/gpurun_out/prof_final_summary.txt
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_final_bench.log | cut -c1-400
head -30 $GRAFT_REPO_ROOT/gpurun_out/prof_final_summary.txt
