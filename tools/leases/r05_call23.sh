#!/bin/bash
O=gpurun_out/r05_call23; mkdir -p $O
timeout 900 python -m pytest tests/test_end_to_end.py -q -m gpu -k pose_refinement --tb=short 2>&1 | grep -v "^iter\|^pretrain" | tail -40 | cut -c1-3000 | tee $O/tests.txt
