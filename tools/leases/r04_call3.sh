#!/bin/bash
# round 4, GPU call 3: lane-ballot masks in the real kernels -- parity, then same-box A/B against the HEAD library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest3.log 2>&1; tail -4 $O/pytest3.log
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export ANERF_LIB=$GRAFT_REPO_ROOT/tools/exp/libanerf_base.so; else unset ANERF_LIB; fi
  timeout 300 python bench.py --workload train --steps 30 --warmup 2 --extra off --cpu-rays 0 > $O/ab_train_${lib}_$rep.json 2>> $O/ab.err
  timeout 300 python bench.py --workload train --n-rand 384 --steps 40 --warmup 2 --extra off --cpu-rays 0 > $O/ab_train384_${lib}_$rep.json 2>> $O/ab.err
done; done
unset ANERF_LIB
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04/ab_train*.json")):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "ERR", e); continue
    ks = {(k["kernel"][:12], k["pass"]): round(k["ms"], 3) for k in j["roofline"]["kernels"]}
    print(f.split("/")[-1], "step median", round(j["step_ms"]["median"], 3), "frac", round(j["roofline"]["frac"], 3), ks)
PY
