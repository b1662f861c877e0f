#!/bin/bash
# round 6, call 32: the one-rank RCCL capture test, repeated (it did not answer once in call 31)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
: > $O/r06_rccl_repeat.txt
for k in 1 2 3 4 5 6; do
  S=$(date +%s)
  timeout 500 python -m pytest tests/test_trajectory.py -q -m gpu -k rccl_collectives 2>&1 | grep -v "^$\|amdgpu.ids" | tail -40 | cut -c1-400 > $O/r06_rccl_try_$k.txt
  E=$(date +%s); echo "try $k: $((E-S)) s: $(tail -1 $O/r06_rccl_try_$k.txt)" >> $O/r06_rccl_repeat.txt
done
cat $O/r06_rccl_repeat.txt
