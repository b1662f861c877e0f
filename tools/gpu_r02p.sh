cd $GRAFT_REPO_ROOT
for v in "SHARED=0 RENDER=0" "SHARED=1 RENDER=0" "SHARED=0 RENDER=1" "SHARED=1 RENDER=1" "SHARED=0 RENDER=0"; do env $v python tools/microbench_train_fwd.py 2>&1 | tail -1; done
for v in "SHARED=0 RENDER=0" "SHARED=0 RENDER=1"; do env $v ANERF_LIB=$GRAFT_REPO_ROOT/tools/exp/libanerf_nosave.so python tools/microbench_train_fwd.py 2>&1 | tail -1; done
