#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c18; mkdir -p $O
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_default.json)"
cp bench_detail.json $O/bench_detail.json
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r05_c18/bench_default.json").read())
print(l["value"], l["roofline"]["frac"], l["clocks"])
print([(e["workload"], e["step_ms"], e.get("sclk_mhz")) for e in l["extras_summary"]])
d=json.load(open("gpurun_out/r05_c18/bench_detail.json"))
print([e.get("clocks") for e in d["extra_workloads"]][1])
PY
