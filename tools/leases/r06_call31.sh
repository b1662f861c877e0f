#!/bin/bash
# round 6, call 31: whole GPU suite + smoke at HEAD (with the sweeps)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
S=$(date +%s)
python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^$" | tail -6 > $O/r06_gpu_suite_head.txt
E=$(date +%s); echo "suite wall $((E-S)) s" >> $O/r06_gpu_suite_head.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_suite_head.txt 2>&1
tail -5 $O/r06_gpu_suite_head.txt | cut -c1-250
