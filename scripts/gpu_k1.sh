#!/bin/bash
# K1 bring-up on the GPU box: engine parity test, full-size drift test, short bench
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wave_row_engine or full_size_known_drift or fused_statistics" > gpurun_out/k1_tests.log 2>&1
rc=$?
tail -15 gpurun_out/k1_tests.log
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/k1_bench.log 2>&1 && tail -2 gpurun_out/k1_bench.log
