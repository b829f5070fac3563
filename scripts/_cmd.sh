cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "warp or field or correct or fp16 or sum" > gpurun_out/e26_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/e26_tests.txt; tail -5 gpurun_out/e26_tests.txt
python scripts/field_warp_time.py 2>&1 | grep -v amdgpu.ids
