#!/bin/bash
# All judged profiles of one build in one gpurun call:  tools/gpu_profiles.sh <round tag, e.g. r02> <git sha>
# -> gpurun_out/<tag>_<workload>_summary.txt (kernel trace + PMC passes) and <tag>_<workload>_bench.json
TAG=$1; SHA=$2
cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/bench.py
# (--graph off: the counters are attributed per kernel dispatch of the eager step -- the same kernels the captured graph replays)
# ONLY="train3072 train384" restricts the run to those workloads (a kernel change that leaves the others untouched)
run() { name=$1; shift; if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi; tools/pmc_passes.sh ${TAG}_$name -- python $B --cpu-rays 0 --extra off --graph off "$@" > /dev/null 2>&1; echo "# build $SHA; command: python bench.py --cpu-rays 0 --extra off --graph off $*" >> gpurun_out/${TAG}_${name}_summary.txt; }
run render64 --steps 3
run api_render64 --workload api_render64 --steps 3
run render64_bf16x3 --precision bf16x3 --steps 3
run hier128_bf16x3 --workload hier128 --precision bf16x3 --steps 2
run train3072 --workload train --steps 10
run train384 --workload train --n-rand 384 --steps 20
run train_mixamo384 --workload train_mixamo --n-rand 384 --opt-pose-step 20 --steps 20
run train_mixamo --workload train_mixamo --steps 10
ls -la gpurun_out/${TAG}_*summary.txt
