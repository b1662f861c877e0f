#!/bin/bash
# round 5, call 18: create_popt on the device + the pose-refinement end-to-end run
O=gpurun_out/r05_call18; mkdir -p $O
timeout 600 python -m pytest tests/test_create_popt.py tests/test_end_to_end.py -x -q -m gpu 2>&1 | tail -5
for cfg in "--from-teacher --pose-noise 0.05 --iters 400" "--from-teacher --pose-noise 0.05 --iters 400 --pose-lrate 0.002 --pose-step 1" "--pose-noise 0.05 --iters 600" "--from-teacher --pose-noise 0.05 --iters 400 --graph off"; do
  echo "== $cfg"
  timeout 600 python tools/train_synthetic.py $cfg 2>&1 | grep -v Saved | tail -4 | cut -c1-1500 | tee -a $O/pose_refine.txt
done
