set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C a-nerf_amd/csrc -j8 > gpurun_out/r02c_make.log 2>&1
timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_fullsize_train.py tests/test_reference_args.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r02c_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02c_pytest.log
B=$GRAFT_REPO_ROOT/bench.py
KT_LINES=8 tools/kt.sh r02c_train3072 -- python $B --workload train --cpu-rays 0 --steps 10 > gpurun_out/r02c_kt_train3072.txt 2>&1
KT_LINES=8 tools/kt.sh r02c_train384 -- python $B --workload train --n-rand 384 --cpu-rays 0 --steps 20 > gpurun_out/r02c_kt_train384.txt 2>&1
python $B --workload train --cpu-rays 0 --steps 20 > gpurun_out/r02c_bench_train3072.json 2>/dev/null
python $B --workload train --n-rand 384 --cpu-rays 0 --steps 40 > gpurun_out/r02c_bench_train384.json 2>/dev/null
tail -3 gpurun_out/r02c_pytest.log
