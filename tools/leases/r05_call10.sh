#!/bin/bash
# round 5, call 10: the captured step on the device timeline (kernel trace of graph replays) next to the eager one
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_c10; mkdir -p $O /tmp/prof
export TMPDIR=/tmp
for g in on off; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/mix384_$g -- python $GRAFT_REPO_ROOT/bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --steps 30 --warmup 3 --extra off --cpu-rays 0 --graph $g --detail /tmp/prof/d_$g.json > /tmp/prof/mix384_$g.log 2>&1); echo "graph $g rc=$?"
  python tools/step_timeline.py /tmp/prof/mix384_$g 22 > $O/mix384_step_timeline_graph_$g.txt 2>&1
  tail -2 $O/mix384_step_timeline_graph_$g.txt
done
head -50 $O/mix384_step_timeline_graph_on.txt
