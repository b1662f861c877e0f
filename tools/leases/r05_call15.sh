#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_trajectory.py -x -q -m gpu -k graphed 2>&1 | tail -5
