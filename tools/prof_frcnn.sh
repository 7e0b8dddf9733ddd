cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pfr
# un-profiled warm run with MIOpen find: fills the user find-db, so the profiled run replays the chosen solvers
timeout -k 10 400 python3 $GRAFT_REPO_ROOT/tools/frcnn_probe.py --find > $GRAFT_REPO_ROOT/gpurun_out/prof_frcnn_warm.log 2>&1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfr -- python3 $GRAFT_REPO_ROOT/tools/frcnn_probe.py --find > $GRAFT_REPO_ROOT/gpurun_out/prof_frcnn.log 2>&1
f=$(ls /tmp/pfr/*/*kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/prof_frcnn_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/summarize_stats.py $f 16 40 > $GRAFT_REPO_ROOT/gpurun_out/prof_frcnn_summary.txt
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_frcnn.log
head -60 $GRAFT_REPO_ROOT/gpurun_out/prof_frcnn_summary.txt | cut -c1-175
