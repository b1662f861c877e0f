cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/bench.py
python $B > gpurun_out/r02n_bench_default.json 2> gpurun_out/r02n_bench_default.err; echo "default rc=$?"
python $B --workload hier --cpu-rays 0 --extra off > gpurun_out/r02n_bench_hier.json 2>/dev/null
python $B --workload hier128 --cpu-rays 0 --extra off > gpurun_out/r02n_bench_hier128.json 2>/dev/null
python $B --workload hier --precision bf16x3 --cpu-rays 0 --extra off > gpurun_out/r02n_bench_hier_b3.json 2>/dev/null
python $B --workload train --precision bf16x3 --cpu-rays 0 --steps 20 > gpurun_out/r02n_bench_train_b3.json 2>/dev/null
python $B --workload train_mixamo --precision bf16x3 --cpu-rays 0 --steps 20 > gpurun_out/r02n_bench_mix_b3.json 2>/dev/null
python $B --workload train --steps 10 > gpurun_out/r02n_bench_train.json 2>/dev/null
python tools/api_bench.py 2>&1 | tail -2
