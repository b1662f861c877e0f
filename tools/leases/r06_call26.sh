#!/bin/bash
# round 6, call 26-27: the seeded configuration sweep (tests/test_hip_sweep.py)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_sweep.py -m gpu -q -s 2>&1 | grep -v "^$" | cut -c1-600 > $O/r06_sweep_first.txt
tail -60 $O/r06_sweep_first.txt
