#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c21; mkdir -p $O
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests_$i.txt 2>&1; echo "pytest $i rc=$?"; tail -2 $O/gpu_tests_$i.txt; done
