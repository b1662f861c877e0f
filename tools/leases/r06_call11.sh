#!/bin/bash
# round 6, call 11: the driver's command (twice), then the judged profiles of build ad143af
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for i in 1 2; do
  /usr/bin/time -f "wall %e s" python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_line_$i.json 2> $O/r06_bench_default_$i.err
  tail -1 $O/r06_bench_default_$i.err; cp bench_detail.json $O/r06_bench_default_$i.json
  python -c "
import json; r=json.loads(open('$O/r06_bench_default_line_$i.json').read().strip().splitlines()[-1]); print(len(json.dumps(r)), r['value'], r['roofline']['frac'], r.get('extras_summary'), r.get('scaling_model_8gpu'))" | cut -c1-2500
done
tools/gpu_profiles.sh r06 ad143af
