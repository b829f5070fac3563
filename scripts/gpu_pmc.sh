#!/bin/bash
# one rocprofv3 --pmc pass over a python script; prints per-kernel mean counter values
# usage: gpu_pmc.sh <tag> "<COUNTER ...>" <script.py> [kernel-name-substring]
set -o pipefail
tag=$1; ctr=$2; prog=$3; pat=${4:-}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/$tag -- python3 $prog > gpurun_out/$tag.log 2>&1
rc=$?
grep -v rocprofv3 gpurun_out/$tag.log | tail -8
f=$(ls -t $(find gpurun_out/$tag -name "*counter_collection.csv") | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] and sys.argv[2] not in r["Kernel_Name"]:
        continue
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size"):
        if k in r:
            acc[r["Kernel_Name"][:60]]["_" + k] = [float(r[k])]
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
exit $rc
