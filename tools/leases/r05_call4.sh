#!/bin/bash
# round 5, call 4: the rest of the GPU suite (call 3 stopped at the ABI-version assert) + the graphed trainer + mixamo384 with pre-captured graphs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/gpu_tests.txt
grep -n "overlap stats\|graphs for keys" $O/gpu_tests.txt
for w in "train_mixamo --n-rand 384 --opt-pose-step 20" "train --n-rand 384"; do
  timeout 300 python bench.py --workload $w --steps 40 --warmup 3 --extra off --cpu-rays 0 --detail $O/d.json > $O/line.json 2>> $O/bench.err
  python - "$w" $O/d.json <<'PY'
import json,sys
j=json.load(open(sys.argv[2]))
print(sys.argv[1], "| step", {k: round(v,4) for k,v in j["step_ms"].items() if k in ("median","p95","max")}, "period", {k: round(v,4) for k,v in j["period_ms"].items() if k in ("median","p95","max")}, "host", round(j["host_enqueue_ms"]["median"],4), "value", round(j["value"]), "graph", j["graph"])
PY
done
