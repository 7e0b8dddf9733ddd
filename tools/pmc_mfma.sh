# MFMA busy cycles per kernel for the train step (eager, few steps: PMC serialises dispatches)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pm
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_]*MFMA[A-Z_0-9]*\|GRBM_GUI_ACTIVE\|SQ_BUSY_CYCLES\|SQ_INSTS_MFMA\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/pmc_counters_available.txt
# un-profiled warm run first: fills MIOpen's user find-db so the counted run replays the chosen solvers only
python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 3 --no-graph --no-micro --no-cpu-baseline > /dev/null 2>&1
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 4 --no-graph --no-micro --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma_bench.log 2>&1
ls /tmp/pm/*/ > $GRAFT_REPO_ROOT/gpurun_out/pmc_ls.txt 2>&1
python3 - <<PY
import csv, glob, collections, os
out = open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_mfma_summary.txt", "w")
f = glob.glob("/tmp/pm/*/*counter_collection.csv")
if not f:
    out.write("no counter_collection.csv\n"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:90]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[k] += 1
tot = collections.defaultdict(float)
for k, d in agg.items():
    for c, v in d.items(): tot[c] += v
out.write("totals: " + repr(dict(tot)) + "\n")
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:40]
for k, d in rows:
    g = d.get("GRBM_GUI_ACTIVE", 0.0); m = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); b = d.get("SQ_BUSY_CYCLES", 0.0)
    out.write(f"{k:90s} calls={calls[k]:5d} gui={g:.3e} mfma_busy={m:.3e} sq_busy={b:.3e} mfma/gui={m/max(g,1):.3f}\n")
out.close()
PY
head -30 $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma_summary.txt | cut -c1-200
