#!/bin/bash
# same-box A/B of bf16x3 kernel variants (box-to-box variance is several %): tools/ab_b3.sh lib1.so lib2.so ...   (in a-nerf_amd/)
R=$(cd "$(dirname "$0")/.." && pwd)
run() { python $R/bench.py "$@" --cpu-rays 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['ms_per_step'],3), end=' ')"; }
for rep in 1 2; do
for lib in "$@"; do
  export ANERF_LIB=$R/a-nerf_amd/$lib
  echo -n "$lib: r64 "; run --precision bf16x3 --steps 5 --warmup 2
  echo -n " hier128 "; run --workload hier128 --precision bf16x3 --steps 2 --warmup 1
  echo -n " train "; run --workload train --precision bf16x3 --steps 30 --warmup 10
  echo
done; done
