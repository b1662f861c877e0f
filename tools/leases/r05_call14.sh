#!/bin/bash
# experiment: prologue kernels as parallel graph branches (ANERF_GRAPH_BRANCHES=1), A/B at the shard size
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c14; mkdir -p $O
for rep in 1 2; do
for br in 0 1; do
  for w in "train_mixamo --n-rand 384 --opt-pose-step 20" "train --n-rand 384"; do
    ANERF_GRAPH_BRANCHES=$br timeout 300 python bench.py --workload $w --steps 60 --warmup 3 --extra off --cpu-rays 0 --graph on --detail $O/d.json > /dev/null 2>> $O/err.txt
    python - "$br" "$w" $O/d.json <<'PY'
import json,sys
j=json.load(open(sys.argv[3]))
print("branches", sys.argv[1], "|", sys.argv[2], "| step median", round(j["step_ms"]["median"],4), "mean", round(j["step_ms"]["mean"],4), "period median", round(j["period_ms"]["median"],4), "loss", j["config"]["loss"])
PY
  done
done
done 2>&1 | tee $O/ab_branches.txt
tail -3 $O/err.txt | cut -c1-300
