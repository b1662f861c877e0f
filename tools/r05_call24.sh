#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_c24 /tmp/prof; export TMPDIR=/tmp
for n in 384 3072; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/cpp_$n -- $GRAFT_REPO_ROOT/tests/csrc/train_step_demo $n 50 > /tmp/prof/cpp_$n.log 2>&1); echo "rc=$?"
  python tools/rocprof_summary.py gpurun_out/r05_c24/train_step_demo_cpp_${n}.txt /tmp/prof/cpp_$n > /dev/null
  echo "# command: rocprofv3 --kernel-trace --stats -- tests/csrc/train_step_demo $n 50   (plain C++ host: 6 + 53 eager iterations, 6 + 50 graph replays)" >> gpurun_out/r05_c24/train_step_demo_cpp_${n}.txt
  head -14 gpurun_out/r05_c24/train_step_demo_cpp_${n}.txt | cut -c1-130
done
