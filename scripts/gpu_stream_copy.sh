#!/bin/bash
# streaming ceiling micro-benchmark: plain run, then rocprofv3 kernel stats and FETCH_SIZE / WRITE_SIZE passes
# usage: scripts/gpu_stream_copy.sh [filter]      (on the GPU box; results under gpurun_out/stream_copy*)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
bin=scripts/ubench/stream_copy
flt=${1:-all}
$bin $flt 10 > gpurun_out/stream_copy_plain.txt 2>&1 || exit 1
cat gpurun_out/stream_copy_plain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stream_copy_stats -- $bin $flt 3 > gpurun_out/stream_copy_stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/stream_copy_$c -- $bin $flt 1 > gpurun_out/stream_copy_$c.log 2>&1 || exit 1
done
python3 scripts/stream_copy_summary.py gpurun_out > gpurun_out/stream_copy_summary.txt
cat gpurun_out/stream_copy_summary.txt
