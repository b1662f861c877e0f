#!/bin/bash
# round 6, call 36: after the watchdog fix -- whole GPU suite twice, smoke, the driver's command, the one-rank RCCL graph bench
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for k in a b; do
  S=$(date +%s)
  python -m pytest tests -x -q -m gpu > $O/r06_gpu_suite_full_$k.log 2>&1
  E=$(date +%s); echo "suite $k wall $((E-S)) s: $(grep -v '^$' $O/r06_gpu_suite_full_$k.log | tail -1)"
  grep -n "^E  \|^FAILED" $O/r06_gpu_suite_full_$k.log | cut -c1-300 | head -20
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_line_5.json 2> $O/r06_bench_default_5.err
E=$(date +%s); echo "driver command wall: $((E - S)) s rc=$?"
cp bench_detail.json $O/r06_bench_default_5.json
python -c "
import json; r=json.loads(open('$O/r06_bench_default_line_5.json').read().strip().splitlines()[-1]); print(len(json.dumps(r)), r['value'], r['roofline']['frac']); print([ (e['workload'], round(e['ms_per_step'],3)) for e in r.get('extras_summary')]); print(r.get('scaling_model_8gpu'))" | cut -c1-1500
ANERF_BENCH_FORCE_DIST=1 python bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph on --steps 200 --warmup 5 --extra off --cpu-rays 0 --detail $O/r06_bench_mixamo384_rccl1_graph_b.json > /tmp/out.json 2> /tmp/err.txt; echo rc=$?
tail -3 /tmp/err.txt | cut -c1-400; cat /tmp/out.json | cut -c1-300
