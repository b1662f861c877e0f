#!/bin/bash
# round 5, call 5: remaining GPU tests, then kernel-trace + PMC passes of every workload on build cea873b
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c5; mkdir -p $O
timeout 900 python -m pytest tests/test_trajectory.py tests/test_variants.py -x -q -m gpu -s > $O/gpu_tests_tail.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/gpu_tests_tail.txt; grep -n "graphs for keys" $O/gpu_tests_tail.txt
bash tools/gpu_profiles.sh r05 cea873b 2>&1 | tail -12
