cd $GRAFT_REPO_ROOT
for tag in base r16; do
  if [ $tag = base ]; then export MCORR_LIB=$PWD/torch_motion_correction_amd/libmcorr.so; else export MCORR_LIB=$PWD/variants/$tag/libmcorr.so; fi
  echo "== $tag"
  bash scripts/gpu_pmc.sh k1w_$tag "WRITE_SIZE" scripts/pipe_probe.py xc_rows_fwd_wave 2>&1 | grep -v "amdgpu.ids\|^[WE]2026" | grep "WRITE_SIZE\|two streams" | head -4
  bash scripts/gpu_prof_py.sh k1t_$tag scripts/pipe_probe.py 2>&1 | grep -v amdgpu.ids | grep "rows_fwd_wave<2, true\|two streams" | head -4
done
