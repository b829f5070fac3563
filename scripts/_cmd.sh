cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "pipeline or movies or sharded or raw or fused" 2>&1 | tail -3
for sch in k1first k1first; do
MC_PIPE_SCHEDULE=$sch timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'warp', r['ms_per_launch'], 'frac', r['frac'], 'solo', r['ms_per_launch_unshared'], 'whole', r['whole_step_frac'], d['config']['shifts_match_ground_truth'])"
done
