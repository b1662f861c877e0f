#!/bin/bash
# round 6, call 8: item 5 pricing (GEMM chain || tile chain), bf16 bypass probe with a whole stage of lead, tests of the touched kernels, timelines
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O /tmp/prof
export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_hip_edge_cases.py tests/test_hip_optim.py tests/test_step_glue.py tests/test_graph_step.py tests/test_fk.py tests/test_trajectory.py tests/test_hip_backward.py -m gpu -q 2>&1 | grep -v "^$" > $O/r06_gpu_tests_d.txt
grep -E "passed|failed|FAILED|Error" $O/r06_gpu_tests_d.txt | cut -c1-300 | tail -12
ANERF_LIB=$GRAFT_REPO_ROOT/tools/exp/libanerf_gemmrows.so timeout 600 python tools/diag/concurrency_probe.py > $O/r06_gemm_tile_concurrency_probe.txt 2>&1
cat $O/r06_gemm_tile_concurrency_probe.txt | cut -c1-250
for w in train_mixamo train; do
  extra=""; [ $w = train_mixamo ] && extra="--opt-pose-step 20"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/${w}384 -- python $GRAFT_REPO_ROOT/bench.py --workload $w --n-rand 384 $extra --steps 30 --warmup 3 --extra off --cpu-rays 0 --graph on --detail /tmp/prof/d_$w.json > /tmp/prof/${w}384.log 2>&1); echo "$w rc=$?"
  python tools/step_timeline.py /tmp/prof/${w}384 22 > $O/r06_${w}384_step_timeline_graph_b.txt 2>&1
  grep -E "k_pack_multi|k_loss|k_reduce_dw2|Fil|# " $O/r06_${w}384_step_timeline_graph_b.txt | cut -c1-110
done
(cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-inline-asm -Wno-unused-result $GRAFT_REPO_ROOT/tools/probe/mfma_probe_bf16.hip -o /tmp/mfma_probe_bf16 2>/dev/null && timeout 300 /tmp/mfma_probe_bf16 > $O/r06_bf16_partial_bypass_probe.txt 2>&1)
grep -E "r6" $O/r06_bf16_partial_bypass_probe.txt | cut -c1-220
