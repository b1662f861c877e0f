#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
ONLY="train_mixamo train_mixamo384" tools/gpu_profiles.sh r06 7bab224
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_line_3.json 2> $O/r06_bench_default_3.err
E=$(date +%s); echo "driver command wall: $((E - S)) s"
cp bench_detail.json $O/r06_bench_default_3.json
python -c "
import json; r=json.loads(open('$O/r06_bench_default_line_3.json').read().strip().splitlines()[-1]); print(len(json.dumps(r)), r['value'], r['roofline']['frac']); print(r.get('extras_summary')); print(r.get('scaling_model_8gpu'))" | cut -c1-3000
