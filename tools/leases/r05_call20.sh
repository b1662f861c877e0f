#!/bin/bash
# round 5, call 20: pose refinement on a subject whose shape follows the pose (analytic ball-and-stick body)
O=gpurun_out/r05_call20; mkdir -p $O
timeout 600 python -m pytest tests/test_create_popt.py -x -q -m gpu 2>&1 | tail -12
for cfg in "--subject spheres --iters 1500" \
           "--subject spheres --pose-noise 0.05 --pretrain 1500 --iters 800 --pose-step 1" \
           "--subject spheres --pose-noise 0.05 --pretrain 1500 --iters 800 --pose-step 1 --net-lrate 0 --pose-coef 0" \
           "--subject spheres --pose-noise 0.05 --iters 2500 --pose-step 4"; do
  echo "== $cfg" | tee -a $O/pose_refine.txt
  timeout 600 python tools/train_synthetic.py $cfg 2>&1 | grep "iter .*00 \|iter     1 \|^{" | cut -c1-1800 | tee -a $O/pose_refine.txt | cut -c1-600
done
