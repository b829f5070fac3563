#!/bin/bash
# Round-end evidence: the default bench line, rocprofv3 kernel stats of the same command,
# the sequential (--no-overlap) variant, and HBM traffic (FETCH_SIZE / WRITE_SIZE passes).
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/round  # gpurun merges old outputs back: start clean
mkdir -p gpurun_out/round
python3 bench.py > gpurun_out/round/bench_default.log 2>&1 || exit 1
tail -1 gpurun_out/round/bench_default.log > gpurun_out/round/bench_line.json
python3 bench.py --no-overlap --no-cpu-baseline --no-secondary > gpurun_out/round/bench_seq.log 2>&1 || exit 1
tail -1 gpurun_out/round/bench_seq.log > gpurun_out/round/bench_line_no_overlap.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/round/stats -- python3 bench.py --no-cpu-baseline --no-secondary > gpurun_out/round/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/round/stats_seq -- python3 bench.py --no-cpu-baseline --no-secondary --no-overlap > gpurun_out/round/stats_seq.log 2>&1 || exit 1
bash scripts/gpu_traffic.sh round/traffic > gpurun_out/round/traffic.txt 2>&1 || exit 1
cat gpurun_out/round/traffic.txt | tail -16
cut -c1-300 gpurun_out/round/bench_line.json
