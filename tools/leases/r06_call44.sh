#!/bin/bash
# round 6, call 44: thread-local captures by default -- the graph tests, the end-to-end run, the driver's command
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_graph_step.py tests/test_trajectory.py tests/test_end_to_end.py tests/test_step_glue.py tests/test_checkpoint.py -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | tail -4 | cut -c1-300
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_line_9.json 2> $O/r06_bench_default_9.err; rc=$?
E=$(date +%s); cp bench_detail.json $O/r06_bench_default_9.json
echo "bench wall $((E-S)) s rc=$rc: $(python -c "
import json; r=json.loads(open('$O/r06_bench_default_line_9.json').read().strip().splitlines()[-1]); m=r['scaling_model_8gpu']; print(len(json.dumps(r)), r['value'], r['roofline']['frac'], [ (e['workload'], round(e['ms_per_step'],3)) for e in r.get('extras_summary')], 'config3', m['config3']['graphed'], m['config3']['eager_overlap'], 'config4', m['config4_opt_pose_step20']['graphed'], m['config4_opt_pose_step20']['eager_overlap'])" 2>&1 | tail -1)"
