cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/bench.py
for lib in a-nerf_amd/libanerf_hip.so tools/exp/libanerf_r01.so a-nerf_amd/libanerf_hip.so; do
  echo "== $lib"
  ANERF_LIB=$GRAFT_REPO_ROOT/$lib python $B --cpu-rays 0 --extra off --steps 6 --precision bf16x3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('render64 b3', round(r['ms_per_step'],2), round(r['roofline']['avg_launch_ms'],2), round(r['roofline']['frac'],4))"
  ANERF_LIB=$GRAFT_REPO_ROOT/$lib python $B --workload train --cpu-rays 0 --steps 20 --precision bf16x3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('train3072 b3', round(r['ms_per_step'],3))"
done
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_backward.py -m gpu -q -x -p no:cacheprovider -k "bf16x3 or b3" 2>&1 | tail -3
