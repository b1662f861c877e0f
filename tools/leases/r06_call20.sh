#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1200 python -X faulthandler -m pytest tests/test_dp_on_device.py tests/test_graph_step.py tests/test_hip_fullsize_train.py tests/test_trajectory.py tests/test_bench_dry_run.py tests/test_step_glue.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | cut -c1-300 | tail -8
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_line_3.json 2> $O/r06_bench_default_3.err
E=$(date +%s); echo "driver command wall: $((E - S)) s"
cp bench_detail.json $O/r06_bench_default_3.json
python -c "
import json; r=json.loads(open('$O/r06_bench_default_line_3.json').read().strip().splitlines()[-1]); print(len(json.dumps(r)), r['value'], r['roofline']['frac']); print(r.get('scaling_model_8gpu'))" | cut -c1-3000
ANERF_BENCH_FORCE_DIST=1 python bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph off --steps 200 --warmup 5 --extra off --cpu-rays 0 --detail $O/r06_bench_mixamo384_rccl1_eager.json 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager overlap, collectives live: step', r.get('step_ms_median'), 'host', r.get('host_enqueue_ms_median'))"
ANERF_BENCH_FORCE_DIST=1 python bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph on --steps 200 --warmup 5 --extra off --cpu-rays 0 --detail $O/r06_bench_mixamo384_rccl1_graph.json 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graphed, collectives live: step', r.get('step_ms_median'), 'host', r.get('host_enqueue_ms_median'), r['config'].get('graph'))"
