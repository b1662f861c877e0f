#!/bin/bash
# round 4, GPU call 1: parity after the ABI-4 change, the driver's exact bench command twice (per-step statistics: where are
# config 4's 34 ms?), the drop-in route alone, and its kernel timeline
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest1.log 2>&1; tail -3 $O/pytest1.log
for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_$i.json 2> $O/bench_default_$i.err; echo "bench $i rc=$?"
done
timeout 300 python bench.py --workload api_render64 --steps 10 --warmup 2 --extra off --cpu-rays 0 > $O/api_render64.json 2> $O/api.err
timeout 300 python bench.py --workload api_render64 --steps 10 --warmup 2 --extra off --cpu-rays 0 --precision bf16x3 > $O/api_render64_b3.json 2>> $O/api.err
mkdir -p /tmp/prof; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/api_kt -- python $GRAFT_REPO_ROOT/bench.py --workload api_render64 --steps 3 --warmup 1 --extra off --cpu-rays 0 > /tmp/prof/api_kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_gaps.py /tmp/prof/api_kt > $GRAFT_REPO_ROOT/$O/api_gaps.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof/mix_kt -- python $GRAFT_REPO_ROOT/bench.py --workload train_mixamo --opt-pose-step 1 --steps 20 --warmup 2 --extra off --cpu-rays 0 > /tmp/prof/mix_kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_gaps.py /tmp/prof/mix_kt > $GRAFT_REPO_ROOT/$O/mix_gaps.txt 2>&1
grep -a '"metric"' /tmp/prof/mix_kt.log > $GRAFT_REPO_ROOT/$O/mix_under_rocprof.json
cd $GRAFT_REPO_ROOT; head -30 $O/api_gaps.txt
