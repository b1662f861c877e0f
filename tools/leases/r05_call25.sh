#!/bin/bash
# round 5, call 25: whole GPU suite + the driver's bench command + smoke on the build with create_popt / from_torch / pose refinement;
# then the joint-from-scratch pose refinement run for 10 000 iterations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c25; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_default.json)"
cp bench_detail.json $O/bench_detail.json
cat $O/bench_default.json
python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python tools/train_synthetic.py --subject spheres --pose-noise 0.05 --iters 10000 --pose-step 4 2>&1 | grep "iter .*[05]00 \|iter     1 \|^{" > $O/pose_refine_joint_10000.txt; tail -3 $O/pose_refine_joint_10000.txt | cut -c1-400
