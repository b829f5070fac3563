#!/bin/bash
# usage: prof_py.sh <tag> <script.py> [pmc groups...]  -> gpurun_out/<tag>/{stats,pmc_i}
set -o pipefail
export TMPDIR=/tmp
tag=$1; script=$2; shift 2
rm -rf gpurun_out/$tag; mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -- python3 $script > gpurun_out/$tag/stats.log 2>&1 || exit 1
i=0
for c in "$@"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/$tag/pmc_$i -- python3 $script > gpurun_out/$tag/pmc_$i.log 2>&1 || exit 1
  i=$((i+1))
done
python3 scripts/prof_summary.py gpurun_out/$tag/stats gpurun_out/$tag/pmc_* > gpurun_out/$tag/summary.txt
cat gpurun_out/$tag/summary.txt
