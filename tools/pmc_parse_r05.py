"""Per-kernel sums of the round-5 PMC passes (tools/prof_r05.sh): HBM traffic per train step per kernel family
((2 FETCH_SIZE + WRITE_SIZE) x 1024 B — FETCH_SIZE reports half of a 16-byte-per-lane stream on gfx950, WRITE_SIZE is exact:
MI355X_MICROARCH.md section HBM) and MFMA utilisation per kernel (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs),
the formula of profiles/r01_i_mfma.md).  The counted program is bench.py's eager train step (7 steps; the first runs MIOpen's
solver search, whose naive_conv_* reference kernels are listed apart)."""
import collections
import csv
import glob
import json
import re
import sys

src, out = sys.argv[1], sys.argv[2]
STEPS = 7


def family(name: str) -> str:
    if re.search(r"bn_(stats|apply|bwd_reduce|bwd_dx)_kernel", name):
        return "fused BN+act"
    if re.search(r"pointwise_(stream|conv|wgrad|bwd|longk)_kernel|pointwise_wgrad32_kernel|conv3x3(s2)?_(fwd|wgrad|tap|wgrad_flat)_kernel|conv3x3s2_(fwd|dgrad|wgrad)_c32_kernel|conv3x3s2_wgrad_sum_kernel|stem6x6_|wgrad_reduce_kernel", name):
        return "hand-written convolutions"
    if "naive_conv" in name:
        return "MIOpen find-mode reference kernels (first step only)"
    if re.search(r"igemm_|ck::|_ZN2ck|SubTensorOp|Cijk_|gemm", name):
        return "library convolutions (MIOpen / CK)"
    if re.search(r"yolo_|upcat|maxpool5|adamw|mosaic|graph_fill|hd_fill", name):
        return "other hand-written"
    return "PyTorch elementwise / copies"


def load(counter_dir, counters):
    f = glob.glob(f"{counter_dir}/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] not in counters:
            continue
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == counters[0]:
            calls[k] += 1
    return agg, calls


fetch, calls = load(f"{src}/pp_FETCH_SIZE", ["FETCH_SIZE"])
write, _ = load(f"{src}/pp_WRITE_SIZE", ["WRITE_SIZE"])
per_kernel, fam = {}, collections.defaultdict(lambda: [0.0, 0])
for k in fetch:
    tb = (2.0 * fetch[k]["FETCH_SIZE"] + write.get(k, {}).get("WRITE_SIZE", 0.0)) * 1024.0
    per_kernel[k] = {"launches": calls[k], "traffic_bytes_per_launch": tb / max(calls[k], 1),
                     "FETCH_KiB": fetch[k]["FETCH_SIZE"], "WRITE_KiB": write.get(k, {}).get("WRITE_SIZE", 0.0)}
    fam[family(k)][0] += tb
    fam[family(k)][1] += calls[k]
res = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only, over `bench.py --steps 3 --warmup 4 "
                "--no-graph` (7 eager YOLOv5s 640^2 bs16 train steps); traffic = (2 FETCH + WRITE) KiB x 1024 (FETCH_SIZE reads half of "
                "a 16-byte-per-lane stream on gfx950; kernels with narrower loads are over-counted by that doubling, at most 2x of "
                "their read bytes); per step = sum over the run / 7",
       "families_per_step": {f: {"traffic_bytes_per_step": int(v[0] / STEPS), "launches_per_step": round(v[1] / STEPS, 1)} for f, v in fam.items()},
       # the 60 largest by bytes per step, plus every kernel bench.py looks up by name whatever its rank
       "kernels": {k[:120]: {"launches_per_step": round(v["launches"] / STEPS, 2), "traffic_bytes_per_launch": int(v["traffic_bytes_per_launch"])}
                   for i, (k, v) in enumerate(sorted(per_kernel.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]))
                   if i < 60 or re.search(r"yolo_|roi_align|pairwise_iou", k)}}
json.dump(res, open(f"{out}/r05_pmc_traffic.json", "w"), indent=1)

mf, mcalls = load(f"{src}/pp_MFMA", ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"])
rows = sorted(mf.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))
with open(f"{out}/r05_pmc_mfma.txt", "w") as o:
    o.write("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); per kernel name, summed over its launches\n")
    tot = collections.defaultdict(lambda: [0.0, 0.0])
    for k, d in rows[:70]:
        g, m = d.get("GRBM_GUI_ACTIVE", 0.0), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        util = m / max(g / 8.0 * 1024.0, 1.0)
        o.write(f"{util:7.3f}  calls={mcalls[k]:5d}  gui={g:.3e}  mfma_busy={m:.3e}  {family(k)[:24]:24s} {k[:110]}\n")
    for k, d in rows:
        f_ = family(k)
        tot[f_][0] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        tot[f_][1] += d.get("GRBM_GUI_ACTIVE", 0.0)
    o.write("\nper family (time-weighted):\n")
    for f_, (m, g) in tot.items():
        o.write(f"  {m / max(g / 8.0 * 1024.0, 1.0):7.3f}  {f_}\n")
print(open(f"{out}/r05_pmc_mfma.txt").read()[:3000])
print(json.dumps(res["families_per_step"], indent=1))
