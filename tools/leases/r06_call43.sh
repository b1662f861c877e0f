#!/bin/bash
# round 6, call 43: k_reduce_dw2 before / after (eighteen partials in flight, 32-bit index arithmetic): kernel-trace averages
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
{
for lib in before after before after; do
  L=""; [ $lib = before ] && L=$GRAFT_REPO_ROOT/tools/exp/libanerf_before_reduce.so
  for wl in "train 1 384" "train_mixamo 20 384" "train 1 3072"; do
    set -- $wl
    echo "== $lib, $1 $3 rays"
    ( [ -n "$L" ] && export ANERF_LIB=$L; KT_LINES=40 bash tools/kt.sh ab_${lib}_$1_$3 -- python $GRAFT_REPO_ROOT/bench.py --workload $1 --n-rand $3 --opt-pose-step $2 --steps 60 --warmup 5 --extra off --cpu-rays 0 --graph on ) | grep "reduce_dw\|k_gemm_tn\|k_adam\|k_pack_multi" | cut -c1-100
  done
done
} > $O/r06_reduce_dw_18_ab.txt 2>&1
cat $O/r06_reduce_dw_18_ab.txt
