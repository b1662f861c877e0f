cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r02j_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02j_pytest.log
tail -3 gpurun_out/r02j_pytest.log
B=$GRAFT_REPO_ROOT/bench.py
for lib in a-nerf_amd/libanerf_hip.so tools/exp/libanerf_r01.so a-nerf_amd/libanerf_hip.so; do
  echo "== $lib"
  ANERF_LIB=$GRAFT_REPO_ROOT/$lib python $B --cpu-rays 0 --extra off --steps 4 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('render64', round(r['ms_per_step'],2), round(r['roofline']['avg_launch_ms'],2), round(r['roofline']['frac'],4))"
  ANERF_LIB=$GRAFT_REPO_ROOT/$lib python tools/microbench_train_fwd.py 2>&1 | tail -1
done
python $B --workload train --cpu-rays 0 --steps 20 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train3072', round(r['ms_per_step'],3), round(r['roofline']['frac'],4))"
python $B --workload train --n-rand 384 --cpu-rays 0 --steps 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train384', round(r['ms_per_step'],3), round(r['roofline']['frac'],4))"
KERNEL_RE="k_mlp_fwd" tools/pmc_one.sh r02j_r64 "FETCH_SIZE" -- python $B --cpu-rays 0 --extra off --steps 2 2>&1 | tail -12
KERNEL_RE="k_mlp_fwd" tools/pmc_one.sh r02j_r64w "WRITE_SIZE" -- python $B --cpu-rays 0 --extra off --steps 2 2>&1 | tail -6
