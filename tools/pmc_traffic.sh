# HBM traffic per launch for the HBM-bound kernels: two counter passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pt_$c -- python3 $GRAFT_REPO_ROOT/tools/traffic_probe.py > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json, os, re
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pt_{c}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        m = re.search(r"(bn_\w+|yolo_decode\w*|mosaic4_\w+|pairwise_iou\w*)", r["Kernel_Name"])
        if m and r["Counter_Name"] == c:
            agg[m.group(1)][0] += float(r["Counter_Value"]); agg[m.group(1)][1] += 1
    for k, (v, n) in agg.items():
        out.setdefault(k, {})[c + "_KiB_per_launch"] = round(v / n, 1)
        out[k]["launches"] = n
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_traffic_more.json", "w"), indent=1)
for k, v in sorted(out.items()): print(k, v)
PY
