cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_optim.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
