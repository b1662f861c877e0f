#!/bin/bash
# round 6, call 28: the seeded sweep 4-fold (48 fp32 + 24 bf16x3 draws) to set its bars
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
ANERF_SWEEP_DRAWS=4 timeout 900 python -m pytest tests/test_hip_sweep.py -m gpu -q -s 2>&1 | grep -v "^$\|amdgpu.ids" | cut -c1-600 > $O/r06_sweep_wide.txt
grep -c "^\.\?F\?seed" $O/r06_sweep_wide.txt; grep "AssertionError\|passed\|failed\|Error" $O/r06_sweep_wide.txt | cut -c1-400 | tail -20
