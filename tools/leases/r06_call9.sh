#!/bin/bash
# round 6, call 9: fused k_mlp_bwd_in + encode backward: parity tests, then A/B (ANERF_NO_FUSED_ENCODE_BWD=1) on config 4 at 3072 / 384 rays
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O /tmp/prof
export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_hip_backward.py tests/test_hip_fullsize_train.py tests/test_variants.py tests/test_trajectory.py tests/test_graph_step.py tests/test_dp_on_device.py tests/test_end_to_end.py -m gpu -q -x 2>&1 | grep -v "^$" > $O/r06_gpu_tests_e.txt
grep -E "passed|failed|FAILED|Error|assert" $O/r06_gpu_tests_e.txt | cut -c1-300 | tail -15
for v in 0 1; do
  for n in 3072 384; do
    ANERF_NO_FUSED_ENCODE_BWD=$v python bench.py --workload train_mixamo --n-rand $n --opt-pose-step 20 --steps 40 --warmup 5 --extra off --cpu-rays 0 --graph on --detail /tmp/prof/d.json 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused' if $v else 'fused  ', $n, 'rays: step_ms median', r.get('step_ms_median'), 'ms_per_step', r['ms_per_step'], 'frac', r['roofline']['frac'])"
  done
done > $O/r06_fused_encode_bwd_ab.txt 2>&1
cat $O/r06_fused_encode_bwd_ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/mix384 -- python $GRAFT_REPO_ROOT/bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --steps 30 --warmup 3 --extra off --cpu-rays 0 --graph on --detail /tmp/prof/d2.json > /tmp/prof/mix384.log 2>&1); echo "rc=$?"
python tools/step_timeline.py /tmp/prof/mix384 22 > $O/r06_train_mixamo384_step_timeline_graph_c.txt 2>&1
cat $O/r06_train_mixamo384_step_timeline_graph_c.txt | cut -c1-110
