#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_fullsize_train.py tests/test_variants.py tests/test_trajectory.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | cut -c1-300 | tail -6
for n in 3072 384; do
  python bench.py --workload train_mixamo --n-rand $n --opt-pose-step 20 --steps 40 --warmup 5 --extra off --cpu-rays 0 --graph on --detail /tmp/d.json 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused+LDS+rsq', $n, 'rays: step_ms median', r.get('step_ms_median'), 'ms_per_step', r['ms_per_step'], 'frac', r['roofline']['frac'])"
done | tee -a $O/r06_fused_encode_bwd_ab.txt
