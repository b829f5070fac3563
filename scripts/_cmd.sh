cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/e9_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/e9_tests.txt
tail -4 gpurun_out/e9_tests.txt
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'warp', r['ms_per_launch'], 'solo', r['ms_per_launch_unshared'], 'whole', r['whole_step_frac'], d['config']['shifts_match_ground_truth'])"
