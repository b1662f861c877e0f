cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_backward.py tests/test_hip_fullsize_train.py tests/test_hip_edge_cases.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25
