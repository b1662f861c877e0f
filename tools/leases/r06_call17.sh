#!/bin/bash
# round 6, call 17: the whole GPU suite + smoke at HEAD (what the driver runs at round end)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
S=$(date +%s)
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -15 > $O/r06_gpu_suite_head.txt
E=$(date +%s); echo "suite wall $((E-S)) s" >> $O/r06_gpu_suite_head.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_suite_head.txt 2>&1
tail -8 $O/r06_gpu_suite_head.txt | cut -c1-250
