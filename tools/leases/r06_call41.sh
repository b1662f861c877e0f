#!/bin/bash
# round 6, call 41: render-sweep outlier 1050 (bf16x3), and the same draw in fp32
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
(timeout 300 python tools/diag/sweep_render_outlier.py 1050 bf16x3; echo; timeout 300 python tools/diag/sweep_render_outlier.py 1050 fp32) 2>&1 | grep -v "amdgpu.ids" | cut -c1-600 > $O/r06_sweep_render_outlier.txt
cat $O/r06_sweep_render_outlier.txt
