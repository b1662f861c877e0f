#!/bin/bash
cd $GRAFT_REPO_ROOT
ONLY="train_mixamo train_mixamo384" tools/gpu_profiles.sh r06 aa477ad
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/mix384 -- python $GRAFT_REPO_ROOT/bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --steps 30 --warmup 3 --extra off --cpu-rays 0 --graph on --detail /tmp/prof/d2.json > /tmp/prof/mix384.log 2>&1); echo "rc=$?"
python tools/step_timeline.py /tmp/prof/mix384 22 > gpurun_out/r06_train_mixamo384_step_timeline_graph_d.txt 2>&1
tail -22 gpurun_out/r06_train_mixamo384_step_timeline_graph_d.txt | cut -c1-100
ANERF_BENCH_FORCE_DIST=1 python bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph off --steps 200 --warmup 5 --extra off --cpu-rays 0 --detail gpurun_out/r06_bench_mixamo384_rccl1_eager.json 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager overlap, collectives live: step', r.get('step_ms_median'), 'host', r.get('host_enqueue_ms_median'))"
