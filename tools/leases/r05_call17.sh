#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/train_synthetic.py --iters 300 --graph on 2>&1 | tail -2 | cut -c1-700
timeout 600 python -m pytest tests/test_end_to_end.py -x -q -m gpu 2>&1 | tail -4
