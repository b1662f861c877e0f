#!/bin/bash
# round 5, call 3: full GPU suite after the coarse-pass split / two early collectives / split Adam; default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/gpu_tests.txt
grep -n "overlap stats" $O/gpu_tests.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
wc -c $O/bench_default.json; cat $O/bench_default.json
cp bench_detail.json $O/bench_detail.json
