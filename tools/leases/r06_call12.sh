#!/bin/bash
# round 6, call 12: LDS-staged bone matrices in k_mlp_bwd_in_enc: parity + timing; then the driver's command
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O /tmp/prof
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_hip_backward.py tests/test_hip_fullsize_train.py tests/test_variants.py tests/test_trajectory.py -m gpu -q 2>&1 | grep -v "^$" > $O/r06_gpu_tests_g.txt
grep -E "passed|failed|FAILED|Error" $O/r06_gpu_tests_g.txt | cut -c1-300 | tail -8
for n in 3072 384; do
  python bench.py --workload train_mixamo --n-rand $n --opt-pose-step 20 --steps 40 --warmup 5 --extra off --cpu-rays 0 --graph on --detail /tmp/prof/d.json 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused+LDS', $n, 'rays: step_ms median', r.get('step_ms_median'), 'ms_per_step', r['ms_per_step'], 'frac', r['roofline']['frac'])"
done >> $O/r06_fused_encode_bwd_ab.txt 2>&1
tail -2 $O/r06_fused_encode_bwd_ab.txt
for i in 1 2; do
  S=$(date +%s)
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_line_$i.json 2> $O/r06_bench_default_$i.err
  E=$(date +%s); echo "driver command wall: $((E - S)) s" | tee -a $O/r06_bench_default_$i.err
  cp bench_detail.json $O/r06_bench_default_$i.json
  python -c "
import json; r=json.loads(open('$O/r06_bench_default_line_$i.json').read().strip().splitlines()[-1]); print(len(json.dumps(r)), r['value'], r['roofline']['frac']); print(r.get('extras_summary')); print(r.get('scaling_model_8gpu'))" | cut -c1-3000
done
