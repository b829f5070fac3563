#!/bin/bash
# Round evidence for BASELINE C3: rocprofv3 kernel stats of the flow, HBM traffic (FETCH_SIZE x 2 /
# WRITE_SIZE, separate passes) and VALU / LDS / wait counters of its kernels -> gpurun_out/c3prof/
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/c3prof; mkdir -p gpurun_out/c3prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3prof/stats -- python3 scripts/c3_flow.py > gpurun_out/c3prof/stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"; do
  tag=$(echo $c | cut -d' ' -f1)
  REPS=2 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/c3prof/pmc_$tag -- python3 scripts/c3_flow.py > gpurun_out/c3prof/pmc_$tag.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, json, os
root = "gpurun_out/c3prof"
stats = max(glob.glob(root + "/stats/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(stats)))
keep = [r for r in rows if not r["Name"].startswith("void at::") and "rocclr" not in r["Name"]][:14]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for r in keep:
    name = r["Name"]
    c = {k: sum(v) / len(v) for k, v in acc.get(name, {}).items()}
    e = {"kernel": name[:90], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    if "FETCH_SIZE" in c: e["read_GB"] = 2 * c["FETCH_SIZE"] * 1024 / 1e9   # gfx950: FETCH_SIZE x 2
    if "WRITE_SIZE" in c: e["write_GB"] = c["WRITE_SIZE"] * 1024 / 1e9
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
              "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE", "SQ_WAVES"):
        if k in c: e[k] = c[k]
    out.append(e)
    print(f"{e['kernel'][:60]:60s} calls={e['calls']:3d} avg_us={e['avg_us']:9.1f} read={e.get('read_GB', 0):6.3f} GB write={e.get('write_GB', 0):6.3f} GB")
json.dump(out, open(root + "/c3_kernels.json", "w"), indent=1)
PY
