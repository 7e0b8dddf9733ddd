# Ordered kernel trace of graph-replayed FasterRCNN 832^2 bs16 steps -> gpurun_out/r03_frcnn_step_order.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && timeout -k 10 820 python3 bench.py --model frcnn --steps 3 --warmup 6 --no-micro --no-cpu-baseline --no-roofline > gpurun_out/frcnn_warm.log 2>&1)
echo "warm rc=$?"
rm -rf /tmp/pfo
(cd $R && timeout -k 10 330 rocprofv3 --kernel-trace --output-format csv -d /tmp/pfo -- python3 bench.py --model frcnn --steps 6 --warmup 6 --no-micro --no-roofline --no-cpu-baseline > gpurun_out/frcnn_order_bench.log 2>&1)
echo "trace rc=$?"
python3 $R/tools/step_order_parse_r03.py /tmp/pfo roi_align_bwd 4 > $R/gpurun_out/r03_frcnn_step_order.txt 2>&1
head -5 $R/gpurun_out/r03_frcnn_step_order.txt
