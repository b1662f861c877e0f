#!/bin/bash
# round 5, call 32: GPU suite + smoke at HEAD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c32; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests.txt
python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
