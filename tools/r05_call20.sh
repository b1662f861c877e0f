#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_trajectory.py -x -q -m gpu -k "resumed" 2>&1 | grep -v "site-packages/_pytest\|pluggy\|runpy" | head -60
