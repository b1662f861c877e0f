#!/bin/bash
# round 5, call 8: split-backward bit test; RCCL (one rank, forced collectives) inside a captured step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c8; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_fullsize_train.py -x -q -m gpu -k "pieces" 2>&1 | tail -5
for g in off on; do
  ANERF_BENCH_FORCE_DIST=1 timeout 200 python bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 4 --steps 12 --warmup 2 --extra off --cpu-rays 0 --graph $g --detail $O/rccl1_$g.json > $O/rccl1_line_$g.json 2> $O/rccl1_$g.err; echo "graph $g rc=$?"
  python - $O/rccl1_$g.json <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    print("loss", j["config"]["loss"], "step median", round(j["step_ms"]["median"],4), "host", round(j["host_enqueue_ms"]["median"],4), "graph", j["graph"], "overlap", j.get("overlap"), "backend", j["backend"], "coll ms", j["collective_ms_per_step"])
except Exception as e:
    print("no record:", e)
PY
  tail -3 $O/rccl1_$g.err | cut -c1-300
done
