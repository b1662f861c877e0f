#!/bin/bash
# round 5, call 30: the end-to-end loop at the reference configs' batch size (N_rand = 3072, surreal.txt / mixamo.txt) -- whole-loop it/s against bench.py's step time
O=gpurun_out/r05_call30; mkdir -p $O
for cfg in "--subject spheres --n-kps 8 --n-cams 6 --hw 128 --n-rand 3072 --n-sample-images 24 --iters 1200" \
           "--subject spheres --n-kps 8 --n-cams 6 --hw 128 --n-rand 3072 --n-sample-images 24 --iters 1200 --pose-noise 0.05 --pose-step 20"; do
  echo "== $cfg" | tee -a $O/e2e_3072_rays.txt
  timeout 800 python tools/train_synthetic.py $cfg 2>&1 | grep "iter .*[02468]00 \|iter     1 \|^{" | tee -a $O/e2e_3072_rays.txt | cut -c1-420
done
