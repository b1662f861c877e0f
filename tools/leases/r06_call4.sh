mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r06_gpu_tests_a.txt
tail -40 gpurun_out/r06_gpu_tests_a.txt | cut -c1-400
