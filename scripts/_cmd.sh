cd $GRAFT_REPO_ROOT
for tag in base w10; do
  if [ $tag = base ]; then export MCORR_LIB=$PWD/torch_motion_correction_amd/libmcorr.so; else export MCORR_LIB=$PWD/variants/$tag/libmcorr.so; fi
  echo "== $tag"
  bash scripts/gpu_prof_py.sh c3_$tag scripts/c3_flow.py 2>&1 | grep -v amdgpu.ids | grep "coherence\|wave1024\|wave512\|xc_rows_inv\|ref_mean" || exit 1
  python scripts/c3_host_profile.py 2>&1 | grep "^estimate\|^correct"
done
export MCORR_LIB=$PWD/variants/w10/libmcorr.so
timeout -k 10 800 python -m pytest tests -x -q -m gpu -k "patch or local or field or prior" > gpurun_out/e32_tests.txt 2>&1; tail -3 gpurun_out/e32_tests.txt
