#!/bin/bash
# rocprofv3 kernel stats of the bench; prints the top kernels.  usage: gpu_prof.sh <tag>
set -o pipefail
tag=${1:-prof}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $BENCH_ARGS > gpurun_out/$tag.log 2>&1
rc=$?
tail -1 gpurun_out/$tag.log | cut -c1-400
f=$(ls -t $(find gpurun_out/$tag -name "*kernel_stats.csv") | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
exit $rc
