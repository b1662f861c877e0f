#!/bin/bash
# round 6, call 38: soak -- the whole GPU suite three times and the driver's bench command three times on one box
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
: > $O/r06_soak_suite_bench.txt
for k in 1 2 3; do
  S=$(date +%s)
  python -m pytest tests -q -m gpu > $O/r06_soak_suite_$k.log 2>&1
  E=$(date +%s); echo "suite $k wall $((E-S)) s: $(grep -v '^$' $O/r06_soak_suite_$k.log | tail -1)" >> $O/r06_soak_suite_bench.txt
  grep -n "^FAILED\|^ERROR" $O/r06_soak_suite_$k.log | cut -c1-300 >> $O/r06_soak_suite_bench.txt
  S=$(date +%s)
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_soak_bench_$k.json 2> $O/r06_soak_bench_$k.err; rc=$?
  E=$(date +%s)
  echo "bench $k wall $((E-S)) s rc=$rc: $(python -c "
import json; r=json.loads(open('$O/r06_soak_bench_$k.json').read().strip().splitlines()[-1]); m=r['scaling_model_8gpu']; print(r['value'], r['roofline']['frac'], 'graphed', m['config3']['graphed'][0], m['config4_opt_pose_step20']['graphed'][0], 'eager', m['config3']['eager_overlap'][0], m['config4_opt_pose_step20']['eager_overlap'][0])" 2>&1 | tail -1)" >> $O/r06_soak_suite_bench.txt
done
cat $O/r06_soak_suite_bench.txt
