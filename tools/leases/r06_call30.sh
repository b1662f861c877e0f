#!/bin/bash
# round 6, call 30: the render-path sweep (tests/test_hip_sweep.py::test_seeded_render_configuration_vs_oracle)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
ANERF_SWEEP_DRAWS=${1:-1} timeout 1200 python -m pytest tests/test_hip_sweep.py -m gpu -q -s -k render 2>&1 | grep -v "^$\|amdgpu.ids" | cut -c1-700 > $O/r06_sweep_render.txt
grep -c "render seed" $O/r06_sweep_render.txt; grep "^E  \|passed\|failed\|^FAILED" $O/r06_sweep_render.txt | grep -v "array\|\[\[" | cut -c1-500 | tail -40
