# Round-5 profiles of the final build (one MI355X).  Writes under gpurun_out/r5/ (copy the summaries to profiles/):
#   r05_bench_command_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command (graph replay + roofline passes)
#   r05_step_trace.txt                   per-step kernel list of the replayed steps
#   r05_pmc_traffic.json, r05_pmc_mfma.txt   per-kernel counter sums over eager train steps (separate --pmc passes, --kernel-trace
#                                        only, the program directly after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
ARGS="--steps 20 --warmup 5 --no-legs --no-cpu-baseline"
(cd $R && python3 bench.py --steps 5 --warmup 6 --no-micro --no-cpu-baseline --no-roofline --no-legs > /tmp/warm.log 2>&1)
rm -rf /tmp/ps1
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps1 -- python3 bench.py $ARGS > gpurun_out/r5/r05_bench_profiled.log 2> gpurun_out/r5/r05_bench_profiled.err)
cp $(ls /tmp/ps1/*/*kernel_stats.csv | head -1) $R/gpurun_out/r5/r05_bench_command_kernel_stats.csv
echo "stats done"
bash $R/tools/step_trace_r05.sh
EAGER="--steps 3 --warmup 4 --no-graph --no-micro --no-cpu-baseline --no-roofline --no-legs"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pp_$c
  (cd $R && timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pp_$c -- python3 bench.py $EAGER > /tmp/pp_$c.log 2>&1)
  echo "pmc $c done"
done
rm -rf /tmp/pp_MFMA
(cd $R && timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pp_MFMA -- python3 bench.py $EAGER > /tmp/pp_MFMA.log 2>&1)
echo "pmc MFMA done"
python3 $R/tools/pmc_parse_r05.py /tmp $R/gpurun_out/r5
