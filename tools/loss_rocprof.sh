cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pl && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -- python3 $GRAFT_REPO_ROOT/tools/loss_probe.py 2>&1 | grep "^loss"; python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("/tmp/pl/*/*kernel_stats.csv")[0])))
for r in rows:
    n=r["Name"]
    if "yolo" in n or "hd_fill" in n:
        import re
        m=re.search(r"(yolo_\w+|hd_fill32_kernel)(<[^>]*>)?", n)
        print(f"{m.group(0):40s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:7.1f}")
PY
