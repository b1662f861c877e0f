#!/bin/bash
# round 6, call 37: test_graph_step.py after the capture_begin-failure handling
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_graph_step.py -q -m gpu -x 2>&1 | grep -v "^$\|amdgpu.ids" | tail -30 | cut -c1-300
