#!/bin/bash
# round 5, call 28: whole GPU suite + the driver's bench command + smoke on the build with create_popt / from_torch / pose refinement;
# then the joint-from-scratch pose refinement run for 10 000 iterations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c28; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_default.json)"
cp bench_detail.json $O/bench_detail.json
cat $O/bench_default.json
python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
