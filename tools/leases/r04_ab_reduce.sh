#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r04/pytest10.log 2>&1; tail -3 gpurun_out/r04/pytest10.log
for r in 1 2; do for lib in base new; do
  if [ $lib = base ]; then export ANERF_LIB=$GRAFT_REPO_ROOT/tools/exp/libanerf_base.so; else unset ANERF_LIB; fi
  for w in "train --n-rand 384" "train_mixamo --n-rand 384 --opt-pose-step 20" "train"; do python bench.py --workload $w --steps 60 --warmup 2 --extra off --cpu-rays 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(\"$lib\", \"$w\", \"median\", round(j[\"step_ms\"][\"median\"],3), \"mean\", round(j[\"step_ms\"][\"mean\"],3))"; done
done; done 2>&1 | tee gpurun_out/r04/ab_reduce.txt
