cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "global or estimate or shifts or pipeline or raw or smoke or patch or k3 or size or rows or spectra or wave" > gpurun_out/e29_tests.txt 2>&1; tail -3 gpurun_out/e29_tests.txt
for s in 1 2; do python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline'].get('whole_step_frac'), d['config'].get('shifts_match_ground_truth'))
"; done
