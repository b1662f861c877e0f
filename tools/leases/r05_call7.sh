#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c7; mkdir -p $O
timeout 600 python tools/train_synthetic.py --iters 300 --graph on > $O/train_synth_on.txt 2>&1; echo "rc=$?"; tail -9 $O/train_synth_on.txt
timeout 600 python -m pytest tests/test_end_to_end.py -x -q -m gpu 2>&1 | tail -15
