# SQ counters of the one-launch 1x1 backward (csrc/pointwise_bwd.hip) next to the dgrad / wgrad pair, on the GPU box:
#   gpurun -- bash tools/bwd_pmc.sh     -> gpurun_out/r4/bwd_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcbwd
(cd $R && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmcbwd -- python3 tools/bwd_pmc.py > /tmp/pmcbwd.log 2>&1)
echo rc=$?
tail -3 /tmp/pmcbwd.log
mkdir -p $R/gpurun_out/r4
python3 - > $R/gpurun_out/r4/bwd_pmc.txt <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcbwd/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if not any(t in k for t in ('pointwise_bwd', 'pointwise_stream', 'pointwise_wgrad')): continue
    key = (k[:100], r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''))
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    n[key] += 1
for k, d in agg.items():
    print(k, 'rows', n[k])
    wc = d.get('SQ_WAVE_CYCLES', 1.0)
    for c, v in sorted(d.items()): print("   %-28s %.4g  (%.1f %% of wave cycles)" % (c, v, 100 * v / wc))
PY
cat $R/gpurun_out/r4/bwd_pmc.txt
# second pass: plain kernel trace (durations) and a pass with the clock / occupancy counters
rm -rf /tmp/pmcbwd2 /tmp/pmcbwd3
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmcbwd2 -- python3 tools/bwd_pmc.py > /tmp/pmcbwd2.log 2>&1)
(cd $R && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmcbwd3 -- python3 tools/bwd_pmc.py > /tmp/pmcbwd3.log 2>&1)
python3 - >> $R/gpurun_out/r4/bwd_pmc.txt <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcbwd2/*/*kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if not any(t in k for t in ('pointwise_bwd', 'pointwise_stream', 'pointwise_wgrad')): continue
    d[k[:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("---- kernel trace durations (us), per launch")
for k, v in d.items(): print(k, ' '.join('%.1f' % t for t in v))
f = glob.glob('/tmp/pmcbwd3/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if not any(t in k for t in ('pointwise_bwd', 'pointwise_stream', 'pointwise_wgrad')): continue
    agg[k[:70]][r['Counter_Name']] += float(r['Counter_Value']); n[k[:70]] += 1
print("---- clock / instruction counters (sums over the launches)")
for k, dd in agg.items():
    print(k, n[k])
    for c, v in sorted(dd.items()): print("   %-22s %.4g" % (c, v))
PY
tail -60 $R/gpurun_out/r4/bwd_pmc.txt
