cd $GRAFT_REPO_ROOT
for tag in base st4; do
  if [ $tag = base ]; then export MCORR_LIB=$PWD/torch_motion_correction_amd/libmcorr.so; else export MCORR_LIB=$PWD/variants/$tag/libmcorr.so; fi
  echo "== $tag"
  bash scripts/gpu_prof_py.sh k1_$tag scripts/k3n_time.py 2>&1 | grep -v amdgpu.ids | grep "global_shifts\|rows_fwd_wave<2, true" || exit 1
done
MCORR_LIB=$PWD/variants/st4s/libmcorr.so python scripts/k3n_time.py 2>&1 | grep "K1 stamps" | tail -1
