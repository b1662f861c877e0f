#!/bin/bash
# round 5, call 21: the new GPU tests (create_popt on the device, pose refinement end to end)
O=gpurun_out/r05_call21; mkdir -p $O
timeout 900 python -m pytest tests/test_create_popt.py tests/test_end_to_end.py -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
