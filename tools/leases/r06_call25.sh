#!/bin/bash
# round 6, call 25: final check at HEAD -- the whole GPU suite, smoke, the driver's command
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
S=$(date +%s)
python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -6 > $O/r06_gpu_suite_head.txt
E=$(date +%s); echo "suite wall $((E-S)) s" >> $O/r06_gpu_suite_head.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $O/r06_gpu_suite_head.txt 2>&1
tail -4 $O/r06_gpu_suite_head.txt | cut -c1-250
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_line_4.json 2> $O/r06_bench_default_4.err
E=$(date +%s); echo "driver command wall: $((E - S)) s"
cp bench_detail.json $O/r06_bench_default_4.json
python -c "
import json; r=json.loads(open('$O/r06_bench_default_line_4.json').read().strip().splitlines()[-1]); print(len(json.dumps(r)), r['value'], r['roofline']['frac']); print([ (e['workload'], round(e['ms_per_step'],3), e.get('host_ms')) for e in r.get('extras_summary')]); print(r.get('scaling_model_8gpu'))" | cut -c1-2500
