#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_graph_step.py -x -q -m gpu 2>&1 | tail -8
