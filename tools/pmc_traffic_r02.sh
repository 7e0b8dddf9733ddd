# HBM traffic per launch for the kernels bench.py's roofline objects name: two counter passes (FETCH_SIZE costs 3 TCC
# slots, WRITE_SIZE 2; never together with a trace domain other than --kernel-trace).  Writes profiles-ready JSON to
# gpurun_out/r02_pmc_traffic.json.  Correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE reports exactly half of a
# wide (16 B per lane) coalesced read stream on gfx950 -> traffic = 2*FETCH + WRITE (float atomics and 16-byte stores are
# counted exactly by WRITE_SIZE).
cd /tmp && export TMPDIR=/tmp
export HD_PROBE_ITERS=3
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt2_$c
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pt2_$c -- python3 $GRAFT_REPO_ROOT/tools/traffic_probe_r02.py > /tmp/pt2_$c.log 2>&1
  tail -2 /tmp/pt2_$c.log
done
python3 - <<PY
import csv, glob, collections, json, os, re
IT = int(os.environ["HD_PROBE_ITERS"])
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pt2_{c}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        n = r["Kernel_Name"]
        m = re.search(r"(bn_stats_kernel|bn_apply_kernel|bn_bwd_reduce_kernel|bn_bwd_dx_kernel|yolo_grad_dense_kernel|pairwise_iou_cols_kernel|pairwise_iou_rows_kernel|roi_align_fwd_cl_kernel|roi_align_bwd_cl_kernel)", n)
        if not m: continue
        key = m.group(1)
        if key.startswith("roi_align_fwd"):
            bf = "bfloat16" in n.split("roi_align_fwd_cl_kernel")[1][:40]
            R = int(r["Grid_Size"]) // int(r["Workgroup_Size"])   # one workgroup per RoI
            key += ("<bf16>" if bf else "<f32>") + f" R={R}"
        agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
    raw[c] = agg
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only (tools/pmc_traffic_r02.sh -> "
                "tools/traffic_probe_r02.py), MI355X, ROCm 7.2.  Counter unit: KiB.  traffic_bytes = (2*FETCH + WRITE)*1024: FETCH_SIZE reads "
                "exactly 1/2 of a 16-byte-per-lane coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact for 16-byte stores "
                "and float atomics.  Infinity-Cache hits are counted by these fabric-side counters, so small re-read tensors still show."}
def tb(key): return (2 * raw["FETCH_SIZE"][key][0] + raw["WRITE_SIZE"][key][0]) * 1024.0
# BN family: all launches of one probe iteration = one train step's BN kernels
bn = ["bn_stats_kernel", "bn_apply_kernel", "bn_bwd_reduce_kernel", "bn_bwd_dx_kernel"]
per = {k: {"launches_per_step": raw["FETCH_SIZE"][k][1] // IT, "FETCH_KiB_per_step": round(raw["FETCH_SIZE"][k][0] / IT, 1),
           "WRITE_KiB_per_step": round(raw["WRITE_SIZE"][k][0] / IT, 1), "traffic_bytes_per_step": int(tb(k) / IT)} for k in bn}
out["bn_family_step"] = {"shape": "YOLOv5s 640 bs16", "launches_per_step": sum(v["launches_per_step"] for v in per.values()),
                         "traffic_bytes": int(sum(tb(k) for k in bn) / IT), "algorithmic_bytes": 6055526400, "per_kernel": per}
def one(key, shape, alg, name=None):
    n = raw["FETCH_SIZE"][key][1]
    if n == 0: return
    out[name or key] = {"shape": shape, "launches": n, "FETCH_SIZE_KiB": round(raw["FETCH_SIZE"][key][0] / n, 1),
                        "WRITE_SIZE_KiB": round(raw["WRITE_SIZE"][key][0] / max(raw["WRITE_SIZE"][key][1], 1), 1),
                        "traffic_bytes": int(tb(key) / n), "algorithmic_bytes": alg}
one("yolo_grad_dense_kernel", "bs16 640 no85", 138700800)
one("pairwise_iou_cols_kernel", "B=2 N=409200 M=500", 2 * (16 * (409200 + 500) + 4 * 409200 * 500))
one("pairwise_iou_rows_kernel", "B=16 N=306900 M=40", 16 * (16 * (306900 + 40) + 4 * 306900 * 40))
maps = sum(16 * 256 * (832 // s) ** 2 for s in (4, 8, 16, 32))
for dn, esz in (("f32", 4), ("bf16", 2)):
    for R in (8192, 2560):
        one(f"roi_align_fwd_cl_kernel<{dn}> R={R}", f"R={R} C=256 P=7 s=2 832 bs16, {dn} maps and output",
            int(R * 256 * 49 * esz + maps * esz))  # upper bound: output + whole pyramid once
one("roi_align_bwd_cl_kernel", "R=2560 C=256 P=7 s=2 (f32 and bf16 incoming gradient, mean), f32 gradient maps",
    int(2560 * 256 * 49 * 3 + 2 * maps * 4))
f2, b2 = out.get("roi_align_fwd_cl_kernel<bf16> R=2560"), out.get("roi_align_bwd_cl_kernel")
if f2 and b2:  # what bench.py --model frcnn reports as `roofline`: one forward + one backward launch of the train step
    out["roi_align_step"] = {"shape": "R=2560 C=256 P=7 s=2 832 bs16 bfloat16", "traffic_bytes": f2["traffic_bytes"] + b2["traffic_bytes"],
                             "algorithmic_bytes_upper_bound": f2["algorithmic_bytes"] + int(2560 * 256 * 49 * 2 + 2 * maps * 4),
                             "note": "algorithmic upper bound = outputs + the whole pyramid once (bench.py counts only the cells under a RoI); random maps and random RoIs of the same shapes as the step's (the step's own RoIs cluster around the ground truth)"}
json.dump(out, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02_pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    if k != "_note": print(k, {a: b for a, b in v.items() if a != "per_kernel"})
PY
