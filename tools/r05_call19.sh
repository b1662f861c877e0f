#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_backward.py -x -q -m gpu -k ragged -s 2>&1 | tail -12
