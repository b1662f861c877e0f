#!/bin/bash
# round 4, GPU call 2: full GPU suite (new: trajectory, drop-in route, dry-run bucket), allocator trace of config 4
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -s > $O/pytest2.log 2>&1; tail -5 $O/pytest2.log
for k in 1 20; do
  ANERF_BENCH_ALLOC_TRACE=1 timeout 300 python bench.py --workload train_mixamo --opt-pose-step $k --steps 40 --warmup 1 --extra off --cpu-rays 0 > $O/mix_alloc_trace_$k.json 2>> $O/mix_alloc.err
done
ANERF_BENCH_ALLOC_TRACE=1 timeout 300 python bench.py --workload train --steps 40 --warmup 1 --extra off --cpu-rays 0 > $O/train_alloc_trace.json 2>> $O/mix_alloc.err
python - <<'PY'
import json
for f in ["mix_alloc_trace_1", "mix_alloc_trace_20", "train_alloc_trace"]:
    try:
        j = json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, "median", round(j["step_ms"]["median"], 3), "max", round(j["step_ms"]["max"], 3), j["allocator_in_timed_region"], j["gc_collections_in_timed_region"])
    for t in j["alloc_trace"]:
        if t["new_segments"] or (t["grew_MB"] or 0) != 0 or t["step_ms"] > 1.1 * j["step_ms"]["median"]:
            print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
PY
