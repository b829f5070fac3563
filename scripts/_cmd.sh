cd $GRAFT_REPO_ROOT
bash scripts/gpu_round_profiles.sh > gpurun_out/round_profiles.log 2>&1; echo "round rc=$?"
tail -20 gpurun_out/round_profiles.log
