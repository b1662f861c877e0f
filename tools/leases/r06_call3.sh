mkdir -p gpurun_out
python tools/diag/fullsize_grad_noise.py > gpurun_out/r06_fullsize_grad_noise_config3.txt 2>&1
python tools/diag/fullsize_grad_noise.py config4 > gpurun_out/r06_fullsize_grad_noise_config4.txt 2>&1
tail -3 gpurun_out/r06_fullsize_grad_noise_config3.txt gpurun_out/r06_fullsize_grad_noise_config4.txt
