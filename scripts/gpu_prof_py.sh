#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python script.  usage: gpu_prof_py.sh <tag> <script.py> [args]
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 "$@" > gpurun_out/$tag.log 2>&1
rc=$?
grep -v rocprofv3 gpurun_out/$tag.log | tail -6
f=$(ls -t $(find gpurun_out/$tag -name "*kernel_stats.csv") | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
exit $rc
