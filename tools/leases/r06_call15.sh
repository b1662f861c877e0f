#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "ANERF_NO_FUSED_ENCODE_BWD=$v"
  ANERF_NO_FUSED_ENCODE_BWD=$v timeout 600 python -m pytest tests/test_hip_backward.py -m gpu -q -s -k "fused_input_gradient" 2>&1 | grep -E "passed|failed|AssertionError|k_mlp_bwd_in_enc<" | cut -c1-250
done
