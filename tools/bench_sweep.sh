#!/bin/bash
# one line per workload: rays/s, ms/step, roofline.frac, dominant-launch ms  (profiles/r02_bench_sweep.txt)
cd "$(dirname "$0")/.."
run() { echo -n "$* : "; python bench.py "$@" --cpu-rays 0 --extra off 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']), round(r['ms_per_step'],3), round(r['roofline']['frac'],4), round(r['roofline'].get('avg_launch_ms',0),3))"; }
run --workload hier --steps 4
run --workload hier128 --steps 3
run --workload hier --precision bf16x3 --steps 4
run --workload hier128 --precision bf16x3 --steps 3
run --workload render64 --precision bf16x3 --steps 5
run --workload train --steps 30 --warmup 10
run --workload train --n-rand 384 --steps 50 --warmup 10
run --workload train --precision bf16x3 --steps 30 --warmup 10
run --workload train_mixamo --steps 30 --warmup 10
run --workload train_mixamo --precision bf16x3 --steps 30 --warmup 10
run --workload train --n-rand 2048 --steps 30 --warmup 10
