cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pbn && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pbn -- python3 $GRAFT_REPO_ROOT/tools/bn_probe.py > /dev/null 2>&1; python3 - <<PY
import csv,glob,collections
rows=list(csv.DictReader(open(glob.glob("/tmp/pbn/*/*kernel_trace.csv")[0])))
import re
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(re.search(r"bn_\w+",r["Kernel_Name"]).group(0), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows if re.search(r"bn_\w+",r["Kernel_Name"])]
# the probe runs, per shape: stats x35, stats x1, apply x35, reduce x35, dx x35  -> print median per consecutive run
import itertools, statistics
out=[]
for k,grp in itertools.groupby(seq, key=lambda t:t[0]):
    v=[d for _,d in grp]
    if len(v)>=10: out.append((k, statistics.median(v)))
for i in range(0,len(out),4): print("  ".join(f"{k}:{v:6.1f}us" for k,v in out[i:i+4]))
PY
