#!/bin/bash
# round 5, call 16: the other workloads of the same build, one line each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c16; mkdir -p $O; : > $O/r05_bench_workloads.jsonl
run() { echo "# python bench.py $* --extra off --cpu-rays 0 (build c7c6d6a)" >> $O/r05_bench_workloads.jsonl; timeout 300 python bench.py "$@" --extra off --cpu-rays 0 --detail $O/d.json 2>/dev/null >> $O/r05_bench_workloads.jsonl; }
run --workload render64x64 --steps 20 --warmup 3
run --workload hier --steps 10 --warmup 2
run --workload hier --steps 10 --warmup 2 --precision bf16x3
run --workload hier128 --steps 5 --warmup 1
run --workload render64 --steps 10 --warmup 2 --precision bf16x3
run --workload api_render64 --steps 10 --warmup 2 --precision bf16x3
run --workload train --n-rand 2048 --steps 20 --warmup 3
run --workload train --steps 20 --warmup 3 --precision bf16x3
run --workload train_mixamo --steps 20 --warmup 3 --precision bf16x3 --opt-pose-step 20
run --workload train --steps 20 --warmup 3 --torch-tail
python - <<'PY'
import json
for l in open("gpurun_out/r05_c16/r05_bench_workloads.jsonl"):
    if l.startswith("{"):
        j=json.loads(l); print(j["config"]["workload"][:70], "|", j["dtype"][:20], "|", round(j["value"]), "rays/s", round(j["ms_per_step"],3), "ms", round(j["roofline"]["frac"],4), j["config"].get("graph"))
PY
