#!/bin/bash
# HBM traffic per kernel of one bench run: separate FETCH_SIZE and WRITE_SIZE passes (MI355X
# guide: FETCH_SIZE x2 on gfx950), joined with kernel durations.  usage: gpu_traffic.sh <tag>
set -o pipefail
tag=${1:-traffic}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_$c -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-overlap > gpurun_out/${tag}_$c.log 2>&1 || exit 1
done
python3 - gpurun_out/${tag}_FETCH_SIZE gpurun_out/${tag}_WRITE_SIZE <<'PY'
import csv, sys, glob, collections, json, os
out = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ns": 0.0})
for d in sys.argv[1:]:
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)  # newest run
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE":
            out[k]["n"] += 1
            out[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = []
for k, v in out.items():
    n = max(v["n"], 1)
    rd, wr, us = 2 * v["FETCH_SIZE"] * 1024 / n, v["WRITE_SIZE"] * 1024 / n, v["ns"] / n / 1e3
    rows.append((us * n, k, n, us, rd / 1e9, wr / 1e9))
rows.sort(reverse=True)
print(f"{'kernel':64s} {'calls':>5s} {'us(pmc)':>8s} {'readGB':>7s} {'writeGB':>7s} {'TB/s':>6s}")
for _, k, n, us, rd, wr in rows[:14]:
    print(f"{k:64s} {n:5d} {us:8.1f} {rd:7.3f} {wr:7.3f} {(rd + wr) / us * 1e3 if us else 0:6.2f}")
json.dump([{"kernel": k, "calls": n, "us_under_pmc": us, "read_GB": rd, "write_GB": wr} for _, k, n, us, rd, wr in rows[:20]],
          open(sys.argv[1] + "/../" + sys.argv[1].split("/")[-1].replace("_FETCH_SIZE", "") + "_summary.json", "w"), indent=1)
PY
