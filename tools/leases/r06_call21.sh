#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
ANERF_BENCH_FORCE_DIST=1 python bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph on --steps 200 --warmup 5 --extra off --cpu-rays 0 --detail $O/r06_bench_mixamo384_rccl1_graph.json > /tmp/out.json 2> /tmp/err.txt; echo rc=$?
tail -5 /tmp/err.txt | cut -c1-600; cat /tmp/out.json | cut -c1-400
