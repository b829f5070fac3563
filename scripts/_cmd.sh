cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "raw or fused or condition" > gpurun_out/e16_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/e16_tests.txt
tail -15 gpurun_out/e16_tests.txt
timeout -k 10 600 python scripts/raw_check.py 2>&1 | tail -5
