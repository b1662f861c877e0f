cd $GRAFT_REPO_ROOT
echo "== new"; python tools/microbench_gemm.py 24576 245760 2>&1 | tail -2
echo "== r01"; ANERF_LIB=$GRAFT_REPO_ROOT/tools/exp/libanerf_r01.so python tools/microbench_gemm.py 24576 245760 2>&1 | tail -2
echo "== new"; python tools/microbench_gemm.py 245760 2>&1 | tail -1
echo "== new b3"; python tools/microbench_gemm.py --b3 245760 2>&1 | tail -1
echo "== r01 b3"; ANERF_LIB=$GRAFT_REPO_ROOT/tools/exp/libanerf_r01.so python tools/microbench_gemm.py --b3 245760 2>&1 | tail -1
timeout 600 python -m pytest tests/test_hip_backward.py tests/test_hip_fullsize_train.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
B=$GRAFT_REPO_ROOT/bench.py
python $B --workload train --cpu-rays 0 --steps 20 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train3072', round(r['ms_per_step'],3), round(r['roofline']['frac'],4))"
python $B --workload train --cpu-rays 0 --steps 20 --precision bf16x3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train3072 b3', round(r['ms_per_step'],3))"
