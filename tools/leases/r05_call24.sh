#!/bin/bash
# round 5, call 24: the pose-refinement record (analytic subject, mixamo.txt's arrangement incl. frame codes)
O=gpurun_out/r05_call24; mkdir -p $O
for cfg in "--subject spheres --pose-noise 0.05 --pretrain 1500 --iters 800 --pose-step 1" \
           "--subject spheres --pose-noise 0.05 --pretrain 1500 --iters 800 --pose-step 1 --graph off" \
           "--subject spheres --pose-noise 0.05 --pretrain 1500 --iters 800 --pose-step 1 --net-lrate 0 --pose-coef 0" \
           "--subject spheres --pose-noise 0.05 --iters 2500 --pose-step 4"; do
  echo "== $cfg" | tee -a $O/pose_refine.txt
  timeout 600 python tools/train_synthetic.py $cfg 2>&1 | grep "iter .*00 \|iter     1 \|^{" | tee -a $O/pose_refine.txt | grep "^{" | cut -c1-200
done
