#!/bin/bash
# round 6, call 27: bf16x3 draws of the seeded sweep on very few samples -- noise or structure?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python tools/diag/sweep_b3_noise.py 12 14 16 17 2>&1 | grep -v amdgpu.ids | cut -c1-700 > $O/r06_sweep_b3_noise.txt
cat $O/r06_sweep_b3_noise.txt
