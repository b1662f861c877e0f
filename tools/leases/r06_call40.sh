#!/bin/bash
# round 6, call 40: both sweeps 10-fold (340 draws), looking for rare failures
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
ANERF_SWEEP_DRAWS=10 timeout 1500 python -m pytest tests/test_hip_sweep.py -m gpu -q -s 2>&1 | grep -v "^$\|amdgpu.ids" | cut -c1-700 > $O/r06_sweep_x10.txt
grep -c "seed [0-9]* \[" $O/r06_sweep_x10.txt; grep "^E  \|passed\|failed\|^FAILED" $O/r06_sweep_x10.txt | grep -v "array\|\[\[" | cut -c1-400 | tail -40
