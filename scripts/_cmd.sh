cd $GRAFT_REPO_ROOT
for tag in base tpf; do
  if [ $tag = base ]; then export MCORR_LIB=$PWD/torch_motion_correction_amd/libmcorr.so; else export MCORR_LIB=$PWD/variants/$tag/libmcorr.so; fi
  echo "== $tag"
  bash scripts/gpu_prof_py.sh c5dose_$tag scripts/c5_dose_one.py 2>&1 | grep -v amdgpu.ids | grep "full_rows\|full_cols\|^[0-9]" | head -5
  bash scripts/gpu_prof_py.sh k3w_$tag scripts/k3_fast_one.py 40 4092 5760 2>&1 | grep -v amdgpu.ids | grep "full_rows\|full_cols" | head -5
done
