cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmcw
(cd $R && timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmcw -- python3 tools/wgrad_pmc.py > /tmp/pmcw.log 2>&1)
echo rc=$?
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcw/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:70] + ' grid ' + r.get('Grid_Size', '?')
    if 'pointwise_wgrad' not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, v))
PY
