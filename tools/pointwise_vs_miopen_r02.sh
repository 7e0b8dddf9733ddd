cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/wt
(cd $R && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/wt -- python3 tools/pointwise_vs_miopen_r02.py > gpurun_out/wg_trace.log 2>&1)
python3 $R/tools/pointwise_vs_miopen_parse_r02.py /tmp/wt > $R/gpurun_out/wg_trace_table.txt 2>&1
cat $R/gpurun_out/wg_trace_table.txt
