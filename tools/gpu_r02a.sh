set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C a-nerf_amd/csrc -j8 > gpurun_out/r02a_make.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/r02a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a_pytest.log
timeout 400 python bench.py > gpurun_out/r02a_bench_default.json 2> gpurun_out/r02a_bench_default.err
tail -c 600 gpurun_out/r02a_bench_default.json
tools/kt.sh r02a_train3072 -- python bench.py --workload train --cpu-rays 0 --steps 10 > gpurun_out/r02a_kt_train3072.txt 2>&1
KT_LINES=14 tools/kt.sh r02a_train384 -- python bench.py --workload train --n-rand 384 --cpu-rays 0 --steps 20 > gpurun_out/r02a_kt_train384.txt 2>&1
tail -5 gpurun_out/r02a_pytest.log
