# HBM traffic per launch of the single kernels bench.py prices (pairwise IoU, RoIAlign forward / gather backward): two counter
# passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; never together with a trace domain other than --kernel-trace), merged
# into gpurun_out/r5/r05_pmc_traffic.json under "single_kernels".  traffic = (2 FETCH + WRITE) x 1024 (MI355X_MICROARCH.md, HBM).
cd /tmp && export TMPDIR=/tmp
export HD_PROBE_ITERS=3
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt4_$c
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pt4_$c -- python3 $GRAFT_REPO_ROOT/tools/traffic_probe_r05.py > /tmp/pt4_$c.log 2>&1
  tail -1 /tmp/pt4_$c.log
done
python3 - <<PY
import csv, glob, collections, json, os, re
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pt4_{c}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        n = r["Kernel_Name"]
        m = re.search(r"(pairwise_iou_cols_kernel|pairwise_iou_rows_kernel|roi_align_fwd_cl_kernel|roi_align_bwd_gather_kernel)", n)
        if not m: continue
        key = m.group(1)
        if key.startswith("roi_align"):
            tail = n.split(key)[1][:60]
            bf = "bfloat16" in tail.split(",")[0]
            grid = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            key += ("<bf16>" if bf else "<f32>") + (f" R={grid}" if "fwd" in key else "")
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
    raw[c] = agg
out = {}
for key in raw["FETCH_SIZE"]:
    n = raw["FETCH_SIZE"][key][1]
    out[key] = {"launches": n, "FETCH_SIZE_KiB": round(raw["FETCH_SIZE"][key][0] / n, 1),
                "WRITE_SIZE_KiB": round(raw["WRITE_SIZE"][key][0] / max(raw["WRITE_SIZE"][key][1], 1), 1),
                "traffic_bytes_per_launch": int((2 * raw["FETCH_SIZE"][key][0] / n + raw["WRITE_SIZE"][key][0] / max(raw["WRITE_SIZE"][key][1], 1)) * 1024)}
p = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5/r05_pmc_traffic.json"
res = json.load(open(p)) if os.path.exists(p) else {}
res["single_kernels"] = out
res["single_kernels_note"] = "tools/pmc_traffic_r05.sh -> tools/traffic_probe_r05.py: bench.py's micro shapes (IoU B=2 N=409200 M=500 and B=16 N=306900 M=40; RoIAlign 832^2 bs16 pyramid C=256 P=7, R=8192 forward, R=2048 forward + gather backward; the backward entry is the mean over its launches of one dtype)"
json.dump(res, open(p, "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
