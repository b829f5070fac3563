cd $GRAFT_REPO_ROOT
for s in k1first all; do echo "== schedule $s"; MC_PIPE_SCHEDULE=$s timeout -k 10 300 python scripts/cumask_probe.py 2>&1 | grep -v amdgpu.ids; done
