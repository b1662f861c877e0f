set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
make -C a-nerf_amd/csrc -j8 > gpurun_out/r02b_make.log 2>&1
timeout 600 python -m pytest tests/test_dp_on_device.py -m gpu -q -p no:cacheprovider > gpurun_out/r02b_pytest_dp.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02b_pytest_dp.log
B=$GRAFT_REPO_ROOT/bench.py
KT_LINES=16 tools/kt.sh r02b_train3072 -- python $B --workload train --cpu-rays 0 --steps 10 > gpurun_out/r02b_kt_train3072.txt 2>&1
KT_LINES=16 tools/kt.sh r02b_train384 -- python $B --workload train --n-rand 384 --cpu-rays 0 --steps 20 > gpurun_out/r02b_kt_train384.txt 2>&1
KT_LINES=20 tools/kt.sh r02b_mix -- python $B --workload train_mixamo --cpu-rays 0 --steps 10 > gpurun_out/r02b_kt_mix.txt 2>&1
tail -5 gpurun_out/r02b_pytest_dp.log
