#!/bin/bash
# round 6, call 29: the sweep's alpha outliers (seeds 40, 43 fp32; 1005, 1019 bf16x3)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for a in "40 fp32" "43 fp32" "1005 bf16x3" "1019 bf16x3"; do timeout 300 python tools/diag/sweep_alpha_outlier.py $a 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|lo, _ =" | cut -c1-700; echo; done > $O/r06_sweep_alpha_outliers.txt
cat $O/r06_sweep_alpha_outliers.txt
