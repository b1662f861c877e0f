#!/bin/bash
# round 6, call 33: whole GPU suite twice (does the one-rank RCCL worker stall again?  its stage log is in the failure text now)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for k in a b; do
  S=$(date +%s)
  python -m pytest tests -x -q -m gpu > $O/r06_gpu_suite_full_$k.log 2>&1
  E=$(date +%s); echo "suite $k wall $((E-S)) s: $(grep -v '^$' $O/r06_gpu_suite_full_$k.log | tail -1)"
  grep -n "did not answer" -A60 $O/r06_gpu_suite_full_$k.log | cut -c1-300 | head -120
done
