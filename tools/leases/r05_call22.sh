#!/bin/bash
# round 5, call 22: pose refinement with mixamo.txt's frame codes on (analytic subject)
O=gpurun_out/r05_call22; mkdir -p $O
timeout 900 python -m pytest tests/test_end_to_end.py -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
for cfg in "--subject spheres --pose-noise 0.05 --pretrain 1500 --iters 800 --pose-step 1" \
           "--subject spheres --pose-noise 0.05 --pretrain 1500 --iters 800 --pose-step 1 --graph off" \
           "--subject spheres --pose-noise 0.05 --pretrain 1500 --iters 800 --pose-step 1 --net-lrate 0 --pose-coef 0"; do
  echo "== $cfg" | tee -a $O/pose_refine.txt
  timeout 600 python tools/train_synthetic.py $cfg 2>&1 | grep "iter .*00 \|iter     1 \|^{" | cut -c1-2000 | tee -a $O/pose_refine.txt | grep -v pretrain | cut -c1-300
done
