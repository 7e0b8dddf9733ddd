# Kernel trace of the graph-replayed YOLOv5s 640^2 bs16 steps (bench.py's timed region) -> gpurun_out/r5/r05_step_trace.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
(cd $R && python3 bench.py --steps 5 --warmup 6 --no-micro --no-cpu-baseline --no-roofline --no-legs > /tmp/warm.log 2>&1)
rm -rf /tmp/pt1
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt1 -- python3 bench.py --steps 30 --warmup 8 --no-micro --no-cpu-baseline --no-roofline --no-legs > gpurun_out/r5/r05_trace_bench.log 2>&1)
python3 $R/tools/step_trace_parse_r02.py /tmp/pt1 200 > $R/gpurun_out/r5/r05_step_trace.txt 2>&1
head -4 $R/gpurun_out/r5/r05_step_trace.txt | cut -c1-150
