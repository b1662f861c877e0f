#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c13; mkdir -p $O
for g in on off; do
  timeout 900 python tools/train_synthetic.py --iters 3000 --graph $g 2>&1 | tail -1 > $O/soak_$g.txt; echo "graph $g rc=$?"; python -c "
import json,sys; j=json.loads(open('$O/soak_$g.txt').read()); print({k:j[k] for k in ('last','it_per_s','host_ms_per_train_batch_median','param_checksum','max_memory_allocated_mb','graphs')})"
done
timeout 600 python -m pytest tests/test_end_to_end.py -x -q -m gpu 2>&1 | tail -3
