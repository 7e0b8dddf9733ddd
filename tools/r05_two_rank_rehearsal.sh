# Round 5: the N = 1 line (headline + packed leg only) and then the 2-rank gloo REHEARSAL of `bench.py --gpus 2` on the one GPU, both
# through the compact stdout line; the second run finds bench_n1_packed.json left by the first and adds eff_vs_packed_n1.
mkdir -p gpurun_out/r5
export HSA_ENABLE_IPC_MODE_LEGACY=0
python - <<'PY' > /dev/null
PY
python bench.py --steps 20 --warmup 8 --no-micro --no-cpu-baseline --leg-budget 60 > gpurun_out/r5/bench_n1_for_rehearsal.log 2> gpurun_out/r5/bench_n1_for_rehearsal.err
tail -c 600 gpurun_out/r5/bench_n1_for_rehearsal.log; echo; cat bench_n1_packed.json; echo
HD_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 10 --warmup 6 --no-micro --no-cpu-baseline --no-legs > gpurun_out/r5/bench_2rank_gloo.log 2> gpurun_out/r5/bench_2rank_gloo.err
echo "rc=$?"; cat gpurun_out/r5/bench_2rank_gloo.log | tail -1 | cut -c1-1800; wc -c gpurun_out/r5/bench_2rank_gloo.log
