set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r02f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02f_pytest.log
B=$GRAFT_REPO_ROOT/bench.py
python $B --workload train --cpu-rays 0 --steps 20 > gpurun_out/r02f_bench_train3072.json 2>/dev/null
python $B --workload train --n-rand 384 --cpu-rays 0 --steps 40 > gpurun_out/r02f_bench_train384.json 2>/dev/null
python $B --workload train_mixamo --cpu-rays 0 --steps 20 > gpurun_out/r02f_bench_mix.json 2>/dev/null
tail -3 gpurun_out/r02f_pytest.log
