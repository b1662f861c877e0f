#!/bin/bash
# round 6, call 34: the watchdog regression test (fixed mode lives, forced global mode dies) + the two one-rank RCCL tests, 4 times
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
: > $O/r06_watchdog_repeat.txt
for k in 1; do
  S=$(date +%s)
  timeout 900 python -m pytest tests/test_graph_step.py tests/test_trajectory.py -q -s -m gpu -k "rccl or watchdog" > $O/r06_watchdog_try_$k.log 2>&1
  E=$(date +%s); echo "try $k: $((E-S)) s: $(grep -v '^$' $O/r06_watchdog_try_$k.log | tail -1)" >> $O/r06_watchdog_repeat.txt
done
cat $O/r06_watchdog_repeat.txt
grep -h "^E  \|Error\|FAILED" $O/r06_watchdog_try_*.log | cut -c1-300 | head -30
grep -h "event queries from" $O/r06_watchdog_try_*.log | cut -c1-400
grep -h "watchdog race" $O/r06_watchdog_try_*.log | cut -c1-500
