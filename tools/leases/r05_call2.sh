#!/bin/bash
# round 5, call 2: the captured training step -- parity tests, then eager vs graph on the shard-size and full-size steps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c2; mkdir -p $O
timeout 900 python -m pytest tests/test_graph_step.py tests/test_abi_exports.py tests/test_hip_optim.py -x -q -m gpu > $O/graph_tests.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/graph_tests.txt
for g in off on; do
  for w in "train --n-rand 384" "train_mixamo --n-rand 384 --opt-pose-step 20" "train" "train_mixamo --opt-pose-step 20"; do
    timeout 300 python bench.py --workload $w --steps 40 --warmup 3 --extra off --cpu-rays 0 --graph $g --detail $O/d.json > $O/line.json 2>> $O/bench.err
    python - "$g" "$w" $O/d.json <<'PY'
import json,sys
j=json.load(open(sys.argv[3]))
print("graph", sys.argv[1], "|", sys.argv[2], "| step_ms median", round(j["step_ms"]["median"],4), "period", round(j["period_ms"]["median"],4), "host_enqueue median", round(j["host_enqueue_ms"]["median"],4), "p95", round(j["host_enqueue_ms"]["p95"],4), "value", round(j["value"]), "frac", round(j["roofline"]["frac"],4), "graph", j["graph"])
PY
  done
done 2>&1 | tee $O/ab_graph.txt
timeout 300 python bench.py --workload render64 --steps 10 --warmup 2 --extra off --cpu-rays 0 --detail $O/d.json | tee $O/render64_line.json
