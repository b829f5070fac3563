cd $GRAFT_REPO_ROOT
for tag in base k2nt base k2nt; do
  if [ $tag = base ]; then export MCORR_LIB=$PWD/torch_motion_correction_amd/libmcorr.so; else export MCORR_LIB=$PWD/variants/$tag/libmcorr.so; fi
  python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$tag', d['ms_per_step'], d['value'], d['roofline'].get('whole_step_frac'))
"; done
export MCORR_LIB=$PWD/variants/k2nt/libmcorr.so
bash scripts/gpu_prof_py.sh k1ps scripts/pipe_probe.py 2>&1 | grep -v amdgpu.ids | grep "rows_fwd_wave<2, true\|xc_cols_fwd\|serial\|two streams" | head -6
