#!/bin/bash
# same-box A/B of fp32 kernel variants: tools/ab_f32.sh lib1.so lib2.so ...   (files in a-nerf_amd/)
R=$(cd "$(dirname "$0")/.." && pwd)
run() { python $R/bench.py "$@" --cpu-rays 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['ms_per_step'],3), round(r['roofline']['avg_launch_ms'],3), end=' ')"; }
for rep in 1 2; do
for lib in "$@"; do
  export ANERF_LIB=$R/a-nerf_amd/$lib
  echo -n "$lib: r64 "; run --precision bf16x3 --steps 0 --warmup 0 > /dev/null 2>&1; run --steps 4 --warmup 1
  echo -n " train "; run --workload train --steps 30 --warmup 10
  echo -n " mix "; run --workload train_mixamo --steps 20 --warmup 5
  echo
done; done
