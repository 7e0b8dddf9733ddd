# Ordered kernel list of ONE graph-replayed YOLOv5s 640^2 bs16 step -> gpurun_out/r5/r05_step_order.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
(cd $R && python3 bench.py --steps 5 --warmup 6 --no-micro --no-cpu-baseline --no-roofline --no-legs > /tmp/warm.log 2>&1)
rm -rf /tmp/po1
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/po1 -- python3 bench.py --steps 12 --warmup 8 --no-micro --no-cpu-baseline --no-roofline --no-legs > /tmp/po1.log 2>&1)
python3 $R/tools/step_order_parse_r03.py /tmp/po1 yolo_grad_dense 5 > $R/gpurun_out/r5/r05_step_order.txt 2>&1
head -3 $R/gpurun_out/r5/r05_step_order.txt | cut -c1-150
