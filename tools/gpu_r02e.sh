cd $GRAFT_REPO_ROOT
for v in g_noload g_nobar g_noasum g_none; do echo "== $v"; ANERF_LIB=$GRAFT_REPO_ROOT/tools/exp/libanerf_$v.so python tools/microbench_gemm.py 245760 2>&1 | tail -1; done
echo "== base"; python tools/microbench_gemm.py 245760 2>&1 | tail -1
