cd $GRAFT_REPO_ROOT
for tag in base k4s; do
  if [ $tag = base ]; then export MCORR_LIB=$PWD/torch_motion_correction_amd/libmcorr.so; else export MCORR_LIB=$PWD/variants/$tag/libmcorr.so; fi
  echo "== $tag"
  bash scripts/gpu_prof_py.sh k4_$tag scripts/pipe_probe.py 2>&1 | grep -v amdgpu.ids | grep "serial\|two streams\|xc_rows_inv" || exit 1
done
