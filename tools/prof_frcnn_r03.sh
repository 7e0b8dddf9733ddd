# BASELINE config 3 (FasterRCNN-R50-PAFPN-DecoupledHead 832^2 bs16): the bench line, then a CLEAN rocprofv3 kernel
# summary of the same command.  The first (un-profiled) run fills MIOpen's user find-db (its solver search for
# ResNet50 at 832^2 takes minutes and runs naive reference kernels); the profiled run then replays the chosen solvers,
# so no naive_conv_* find kernels pollute the trace.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
STEPS=${STEPS:-10}
WARM=${WARM:-6}
(cd $R && timeout -k 10 820 python3 bench.py --model frcnn --steps $STEPS --warmup $WARM --no-micro > gpurun_out/r03_frcnn_bench.log 2> gpurun_out/r03_frcnn_bench.err)
echo "bench rc=$?"; tail -2 $R/gpurun_out/r03_frcnn_bench.err | cut -c1-300
rm -rf /tmp/pfr3
(cd $R && timeout -k 10 330 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfr3 -- python3 bench.py --model frcnn --steps $STEPS --warmup $WARM --no-micro --no-roofline --no-cpu-baseline > gpurun_out/r03_frcnn_bench_profiled.log 2> gpurun_out/r03_frcnn_bench_profiled.err)
echo "profiled rc=$?"
f=$(ls /tmp/pfr3/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/r03_frcnn_kernel_stats.csv
python3 $R/tools/summarize_stats.py $f $((STEPS + WARM + 2)) 40 > $R/gpurun_out/r03_frcnn_summary.txt
grep -c naive_conv $f
head -24 $R/gpurun_out/r03_frcnn_summary.txt | cut -c1-170
