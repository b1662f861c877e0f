cd $GRAFT_REPO_ROOT
for lib in tools/exp/libanerf_ns_save.so tools/exp/libanerf_ns_nosave.so tools/exp/libanerf_ns_noglds_save.so tools/exp/libanerf_ns_noglds_nosave.so tools/exp/libanerf_ns_save.so; do
  ANERF_LIB=$GRAFT_REPO_ROOT/$lib python tools/microbench_train_fwd.py 2>&1 | tail -1
done | tee gpurun_out/r02d6_ablate_train_fwd.txt
