cd $GRAFT_REPO_ROOT
bash scripts/gpu_prof_py.sh c5dose scripts/c5_dose_one.py 2>&1 | grep -v amdgpu.ids | head -14
