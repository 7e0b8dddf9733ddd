# rocprofv3 kernel summaries for the headline workload (YOLOv5s 640^2 bs16 bf16):
#   (1) the train steps alone (no roofline / micro legs): per-step breakdown by kernel family;
#   (2) the SAME COMMAND the driver runs (python bench.py, minus the CPU baseline leg, which launches no kernels):
#       the average duration of the roofline kernels in this trace is what bench.py's HIP-event numbers must agree with.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# warm MIOpen's find-db un-profiled, so the profiled runs replay the chosen solvers
(cd $R && python3 bench.py --steps 5 --warmup 6 --no-micro --no-cpu-baseline --no-roofline > /tmp/warm.log 2>&1)
rm -rf /tmp/pf1 /tmp/pf2
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf1 -- python3 bench.py --steps 40 --warmup 8 --no-micro --no-cpu-baseline --no-roofline > gpurun_out/r02_prof_steps_bench.log 2>&1)
f=$(ls /tmp/pf1/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r02_train_step_kernel_stats.csv
python3 $R/tools/summarize_stats.py $f 48 40 > $R/gpurun_out/r02_train_step_summary.txt
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf2 -- python3 bench.py --no-cpu-baseline > gpurun_out/r02_prof_full_bench.log 2> gpurun_out/r02_prof_full_bench.err)
f=$(ls /tmp/pf2/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r02_bench_command_kernel_stats.csv
head -22 $R/gpurun_out/r02_train_step_summary.txt | cut -c1-160
grep "bn_\|yolo_grad_dense\|pairwise_iou" $R/gpurun_out/r02_bench_command_kernel_stats.csv | cut -c1-60,200-400 | head -12
