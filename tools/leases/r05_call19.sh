#!/bin/bash
# round 5, call 19: pose refinement in isolation (frozen teacher networks, regulariser off / on)
O=gpurun_out/r05_call19; mkdir -p $O
timeout 600 python -m pytest tests/test_create_popt.py -x -q -m gpu 2>&1 | tail -3
for cfg in "--from-teacher --net-lrate 0 --pose-coef 0 --pose-step 1 --pose-noise 0.03 --iters 400" \
           "--from-teacher --net-lrate 0 --pose-coef 0 --pose-step 1 --pose-noise 0.03 --iters 400 --n-rand 4096 --n-sample-images 16" \
           "--from-teacher --net-lrate 0 --pose-step 1 --pose-noise 0.03 --iters 400" \
           "--from-teacher --net-lrate 0 --pose-coef 0 --pose-step 4 --pose-noise 0.03 --iters 800 --pose-lrate 0.0002"; do
  echo "== $cfg" | tee -a $O/pose_refine.txt
  timeout 600 python tools/train_synthetic.py $cfg 2>&1 | grep "^iter\|^{" | cut -c1-1500 | tee -a $O/pose_refine.txt | cut -c1-400
done
