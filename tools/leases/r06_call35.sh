#!/bin/bash
# round 6, call 35: which event queries are refused during a capture (rules of the runtime)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 300 python tools/diag/capture_event_query_rules.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 > $O/r06_capture_event_query_rules.txt
cat $O/r06_capture_event_query_rules.txt
