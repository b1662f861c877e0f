#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -X faulthandler -m pytest tests/test_trajectory.py -m gpu -q -k "rccl" 2>&1 | tail -15 | cut -c1-400
