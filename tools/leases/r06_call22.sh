#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_dp_on_device.py tests/test_graph_step.py tests/test_hip_optim.py tests/test_step_glue.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | cut -c1-300 | tail -6
for w in "train_mixamo --opt-pose-step 20" "train"; do
ANERF_BENCH_FORCE_DIST=1 MASTER_PORT=$((29600 + RANDOM % 300)) python bench.py --workload $w --n-rand 384 --graph off --steps 200 --warmup 5 --extra off --cpu-rays 0 --detail /tmp/d.json 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager overlap, collectives live, $w: step', r.get('step_ms_median'), 'host', r.get('host_enqueue_ms_median'))"
done
ANERF_PROFILE_SINGLE_THREAD=1 ANERF_BENCH_FORCE_DIST=1 MASTER_PORT=$((29600 + RANDOM % 300)) python tools/host_profile.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph off --steps 300 --warmup 5 --extra off --cpu-rays 0 > /dev/null 2> $O/r06_host_profile_eager_overlap_single_thread_b.txt
grep -A42 "Ordered by" $O/r06_host_profile_eager_overlap_single_thread_b.txt | cut -c1-150 | tail -38
