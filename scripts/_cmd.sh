cd $GRAFT_REPO_ROOT
for tag in base r16 base r16; do
  if [ $tag = base ]; then export MCORR_LIB=$PWD/torch_motion_correction_amd/libmcorr.so; else export MCORR_LIB=$PWD/variants/$tag/libmcorr.so; fi
  for st in 20 60; do
  python bench.py --steps $st --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', $st, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('whole_step_frac'))
"; done; done
