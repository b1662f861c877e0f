cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/bench.py
KT_LINES=30 tools/kt.sh r02i_train3072 -- python $B --workload train --cpu-rays 0 --steps 10 > gpurun_out/r02i_kt_train3072.txt 2>&1
KT_LINES=30 tools/kt.sh r02i_train384 -- python $B --workload train --n-rand 384 --cpu-rays 0 --steps 20 > gpurun_out/r02i_kt_train384.txt 2>&1
