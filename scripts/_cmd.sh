cd $GRAFT_REPO_ROOT
for tag in base nw5; do
  if [ $tag = base ]; then export MCORR_LIB=$PWD/torch_motion_correction_amd/libmcorr.so; else export MCORR_LIB=$PWD/variants/$tag/libmcorr.so; fi
  echo "== $tag"
  bash scripts/gpu_prof_py.sh c3_$tag scripts/c3_flow.py 2>&1 | grep -v amdgpu.ids | grep "near_wave1024" || exit 1
  bash scripts/gpu_prof_py.sh k3n_$tag scripts/k3n_time.py 2>&1 | grep -v amdgpu.ids | grep "cols_inv_near" || exit 1
done
