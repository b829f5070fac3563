cd $GRAFT_REPO_ROOT
bash scripts/gpu_round_profiles.sh > gpurun_out/round_profiles.log 2>&1 || { tail -20 gpurun_out/round_profiles.log; exit 1; }
tail -20 gpurun_out/round_profiles.log | cut -c1-400
bash scripts/gpu_c3_profiles.sh > gpurun_out/c3_profiles.log 2>&1 || { tail -20 gpurun_out/c3_profiles.log; exit 1; }
tail -16 gpurun_out/c3_profiles.log
