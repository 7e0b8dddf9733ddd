#!/bin/bash
# the round's last GPU call: counters / traces of the final build, then the full GPU suite and the default bench
bash tools/r05_profiles.sh > gpurun_out/r5_final_profiles.out 2>&1
tail -3 gpurun_out/r5_final_profiles.out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5/smoke.log 2>&1; tail -1 gpurun_out/r5/smoke.log
bash tools/r05_suite_bench.sh
