#!/bin/bash
# round 5, call 11: soak -- 3000 iterations end to end, captured vs eager: same final numbers, flat memory
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c11; mkdir -p $O
for g in on off; do
  timeout 900 python tools/train_synthetic.py --iters 3000 --graph $g 2>&1 | grep -v "^iter .*[1-9]50 \|Saved" | tail -8 > $O/soak_$g.txt; echo "graph $g rc=$?"; tail -2 $O/soak_$g.txt | cut -c1-900
done
