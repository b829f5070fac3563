cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['whole_step_frac'])
print(d['raw_u8'])
"
