#!/bin/bash
# usage: prof_args.sh <tag> <pmc groups separated by ;> <script.py> [args]
set -o pipefail
export TMPDIR=/tmp
tag=$1; pmcs=$2; shift 2
rm -rf gpurun_out/$tag; mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -- python3 "$@" > gpurun_out/$tag/stats.log 2>&1 || exit 1
i=0
IFS=';' read -ra GR <<< "$pmcs"
for c in "${GR[@]}"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/$tag/pmc_$i -- python3 "$@" > gpurun_out/$tag/pmc_$i.log 2>&1 || exit 1
  i=$((i+1))
done
python3 scripts/prof_summary.py gpurun_out/$tag/stats gpurun_out/$tag/pmc_* > gpurun_out/$tag/summary.txt
cat gpurun_out/$tag/summary.txt
