#!/bin/bash
# round 5, call 26: the reader's staged (pinned arena, one asynchronous copy) upload against pageable copies, end-to-end iterations per second
O=gpurun_out/r05_call26; mkdir -p $O
for cfg in "--iters 1500 --staged-uploads off" "--iters 1500 --staged-uploads on" \
           "--subject spheres --pose-noise 0.05 --iters 1500 --pose-step 4 --staged-uploads off" \
           "--subject spheres --pose-noise 0.05 --iters 1500 --pose-step 4 --staged-uploads on" \
           "--iters 1500 --staged-uploads on --graph off"; do
  echo "== $cfg" | tee -a $O/staged_uploads_ab.txt
  timeout 600 python tools/train_synthetic.py $cfg 2>&1 | grep "^{" | tee -a $O/staged_uploads_ab.txt | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(round(d['it_per_s'],1),'it/s host', round(d['host_ms_per_train_batch_median'],3), 'last', d['last'], d['param_checksum'], d['pose_refinement'] and d['pose_refinement']['mpjpe_mm_end'])"
done
