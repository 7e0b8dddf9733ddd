mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
HD_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 6 --no-micro --no-cpu-baseline --no-legs > gpurun_out/r5/bench_2rank_gloo.log 2> gpurun_out/r5/bench_2rank_gloo.err
echo "rc=$?"; tail -40 gpurun_out/r5/bench_2rank_gloo.err | cut -c1-180
