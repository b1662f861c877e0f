mkdir -p gpurun_out
python tools/diag/fullsize_grad_noise.py > gpurun_out/r06_fullsize_grad_noise_config3.txt 2>&1
python tools/diag/fullsize_grad_noise.py config4 > gpurun_out/r06_fullsize_grad_noise_config4.txt 2>&1
tail -3 gpurun_out/r06_fullsize_grad_noise_config3.txt gpurun_out/r06_fullsize_grad_noise_config4.txt
ANERF_BENCH_FORCE_DIST=1 python tools/host_profile.py --workload train_mixamo --n-rand 384 --opt-pose-step 20 --graph off --steps 300 --warmup 5 --extra off --cpu-rays 0 > gpurun_out/r06_host_profile_eager_overlap.json 2> gpurun_out/r06_host_profile_eager_overlap.txt
tail -70 gpurun_out/r06_host_profile_eager_overlap.txt | cut -c1-180
