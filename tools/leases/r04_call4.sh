#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest4.log 2>&1; tail -4 $O/pytest4.log
for rep in 1 2; do
  timeout 300 python bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --steps 60 --warmup 2 --extra off --cpu-rays 0 > $O/mix384_fusedpose_$rep.json 2>> $O/c4.err
  timeout 300 python bench.py --workload train_mixamo --opt-pose-step 20 --steps 30 --warmup 2 --extra off --cpu-rays 0 > $O/mix3072_fusedpose_$rep.json 2>> $O/c4.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04/mix*_fusedpose_*.json")):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "ERR", e); continue
    print(f.split("/")[-1], "step median", round(j["step_ms"]["median"], 3), "host", round(j["host_enqueue_ms"]["median"], 3), "frac", round(j["roofline"]["frac"], 3), "mfma sum", round(sum(k["ms"] for k in j["roofline"]["kernels"]), 3))
PY
