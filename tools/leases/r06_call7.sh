#!/bin/bash
# round 6, call 7: tests of the fused small kernels + failed-capture cleanup, step timelines (graph on), bf16 partial-bypass probe
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O /tmp/prof
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_graph_step.py tests/test_hip_edge_cases.py tests/test_fk.py tests/test_trajectory.py tests/test_hip_backward.py tests/test_hip_fullsize_train.py tests/test_dp_on_device.py tests/test_hip_optim.py tests/test_dropin_route.py -m gpu -q -s 2>&1 | grep -v "^$" > $O/r06_gpu_tests_c.txt
grep -E "passed|failed|FAILED|Error|full-size" $O/r06_gpu_tests_c.txt | cut -c1-400 | tail -25
for w in train_mixamo train; do
  extra=""; [ $w = train_mixamo ] && extra="--opt-pose-step 20"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/${w}384 -- python $GRAFT_REPO_ROOT/bench.py --workload $w --n-rand 384 $extra --steps 30 --warmup 3 --extra off --cpu-rays 0 --graph on --detail /tmp/prof/d_$w.json > /tmp/prof/${w}384.log 2>&1); echo "$w rc=$?"
  python tools/step_timeline.py /tmp/prof/${w}384 22 > $O/r06_${w}384_step_timeline_graph_a.txt 2>&1
  tail -1 $O/r06_${w}384_step_timeline_graph_a.txt
done
cat $O/r06_train_mixamo384_step_timeline_graph_a.txt | cut -c1-120
(cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-inline-asm -Wno-unused-result $GRAFT_REPO_ROOT/tools/probe/mfma_probe_bf16.hip -o /tmp/mfma_probe_bf16 2>/dev/null && timeout 300 /tmp/mfma_probe_bf16 > $O/r06_bf16_partial_bypass_probe.txt 2>&1)
grep -E "r6|MFMA only|NO MFMA" $O/r06_bf16_partial_bypass_probe.txt | cut -c1-220
