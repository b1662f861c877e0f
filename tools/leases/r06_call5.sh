mkdir -p gpurun_out
python -m pytest tests/test_graph_step.py tests/test_step_glue.py tests/test_bench_dry_run.py tests/test_hip_fullsize_train.py -m gpu -q -s -k "failing_capture or failed_capture or launches_its_own or torch_distributed_run or real_flat_bucket or chunked_oracle or rccl" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r06_gpu_tests_b.txt
tail -30 gpurun_out/r06_gpu_tests_b.txt | cut -c1-600
ANERF_BENCH_TORCH_PROFILE=1 python bench.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph off --steps 5 --warmup 2 --extra off --cpu-rays 0 > /dev/null 2> gpurun_out/r06_torch_profile_mix384.txt
grep -n "^aten::" gpurun_out/r06_torch_profile_mix384.txt | cut -c1-700 | head -40
