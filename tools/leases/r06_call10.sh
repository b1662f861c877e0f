#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1500 python -X faulthandler -m pytest tests/test_hip_backward.py tests/test_hip_fullsize_train.py tests/test_variants.py tests/test_trajectory.py tests/test_graph_step.py tests/test_dp_on_device.py tests/test_end_to_end.py tests/test_hip_parity.py -m gpu -q -s 2>&1 | grep -v "^$" > $O/r06_gpu_tests_f.txt
grep -E "passed|failed|FAILED|Error|full-size|observed" $O/r06_gpu_tests_f.txt | cut -c1-330 | tail -30
