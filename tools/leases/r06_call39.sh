#!/bin/bash
# round 6, call 39: the driver's command three times after the longer warm-up of the scaling model's live runs
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for k in 6 7 8; do
  S=$(date +%s)
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_line_$k.json 2> $O/r06_bench_default_$k.err; rc=$?
  E=$(date +%s); cp bench_detail.json $O/r06_bench_default_$k.json
  echo "bench $k wall $((E-S)) s rc=$rc: $(python -c "
import json; r=json.loads(open('$O/r06_bench_default_line_$k.json').read().strip().splitlines()[-1]); m=r['scaling_model_8gpu']; print(len(json.dumps(r)), r['value'], r['roofline']['frac'], 'config3', m['config3']['graphed'], m['config3']['eager_overlap'], 'config4', m['config4_opt_pose_step20']['graphed'], m['config4_opt_pose_step20']['eager_overlap'])" 2>&1 | tail -1)"
done
