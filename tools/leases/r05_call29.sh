#!/bin/bash
# round 5, call 29: joint pose refinement with more views (8 poses x 6 cameras instead of 4 x 4): does the plateau move?
O=gpurun_out/r05_call29; mkdir -p $O
for cfg in "--subject spheres --pose-noise 0.05 --n-kps 8 --n-cams 6 --n-sample-images 16 --iters 8000 --pose-step 4"; do
  echo "== $cfg" | tee -a $O/pose_refine_48_images.txt
  timeout 800 python tools/train_synthetic.py $cfg 2>&1 | grep "iter .*000 \|iter     1 \|iter .*500 \|^{" | tee -a $O/pose_refine_48_images.txt | cut -c1-300
done
