mkdir -p gpurun_out/r5
bash tools/prof_r05.sh > gpurun_out/r5/prof_r05.out 2>&1; tail -30 gpurun_out/r5/prof_r05.out
bash tools/step_order_r05.sh >> gpurun_out/r5/prof_r05.out 2>&1
bash tools/pmc_traffic_r05.sh >> gpurun_out/r5/prof_r05.out 2>&1; tail -8 gpurun_out/r5/prof_r05.out
ls -la gpurun_out/r5 | tail -20
