#!/bin/bash
# round 5, call 6: the whole GPU suite + the driver's bench command twice on the final build
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c6; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/gpu_tests.txt
for i in 1 2; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_$i.json 2> $O/bench_default_$i.err; echo "bench $i rc=$? bytes=$(wc -c < $O/bench_default_$i.json)"
  cp bench_detail.json $O/bench_detail_$i.json
done
cat $O/bench_default_2.json
python -c "
import __graft_entry__ as g; g.smoke()"
