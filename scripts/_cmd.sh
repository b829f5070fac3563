cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "raw or fused or pipeline" 2>&1 | tail -8
