#!/bin/bash
# round 4, final measurement call: judged profiles of build d3bc96e (kernel trace + PMC passes per workload) and the driver's command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tools/gpu_profiles.sh r04 d3bc96e
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "default rc=$?"
for w in "render64" "render64 --precision bf16x3" "hier" "hier128" "hier128 --precision bf16x3" "train" "train --n-rand 384" "train --n-rand 2048" "train --precision bf16x3" "train_mixamo --opt-pose-step 20" "train_mixamo --precision bf16x3" "train_mixamo --n-rand 384 --opt-pose-step 20" "api_render64" "api_render64 --precision bf16x3"; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --extra off --cpu-rays 0 2>/dev/null | sed "s|^|# python bench.py --workload $w --steps 20 --warmup 3 --extra off --cpu-rays 0 (build d3bc96e)\n|" >> gpurun_out/r04_bench_workloads.jsonl
done
python tools/microbench_mlp.py --pre > gpurun_out/r04_microbench_mlp_pre.txt 2>&1
ls -la gpurun_out/r04_*
