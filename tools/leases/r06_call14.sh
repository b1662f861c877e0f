#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_backward.py -m gpu -q -s -k "fused_input_gradient or one_call_training_step" 2>&1 | grep -E "passed|failed|FAILED|Error|k_mlp_bwd_in_enc<" | cut -c1-250
for lib in "" tools/exp/libanerf_enc_nostore.so; do
  for n in 3072 384; do
    ANERF_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python bench.py --workload train_mixamo --n-rand $n --opt-pose-step 20 --steps 40 --warmup 5 --extra off --cpu-rays 0 --graph on --detail /tmp/d.json 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-product}', $n, 'rays: step_ms median', r.get('step_ms_median'))"
  done
done
ANERF_PROFILE_SINGLE_THREAD=1 ANERF_BENCH_FORCE_DIST=1 python tools/host_profile.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph off --steps 300 --warmup 5 --extra off --cpu-rays 0 > /dev/null 2> $O/r06_host_profile_eager_overlap_single_thread.txt
grep -A60 "Ordered by" $O/r06_host_profile_eager_overlap_single_thread.txt | cut -c1-160
